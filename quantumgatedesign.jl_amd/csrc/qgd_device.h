// qgd_device.h -- launch context shared by the kernel translation units (qgd_k_*.hip) and the host side (qgd_host_*.cpp)
#ifndef QGD_DEVICE_H
#define QGD_DEVICE_H
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#define QGD_MAX_OPS_DEV 8

/* Path overrides, for tests and A/B timing only: QGD_PATHS="token[=value],token,..." forces a kernel selection the library
   otherwise makes from the problem's shape (INTEGRATION.md section 5 lists the tokens).  Every token names a path the
   library takes by itself for some shape -- variants that lost a measurement are not kept behind switches.  Read at every
   call: tests change it between two handles of one process.  Returns the token's value ("" when it has none) or NULL. */
#include <stdlib.h>
#include <string.h>
static inline const char *qgd_path(const char *key)
{
    const char *e = getenv("QGD_PATHS");
    const size_t n = strlen(key);
    while (e && *e) {
        if (strncmp(e, key, n) == 0 && (e[n] == '\0' || e[n] == ',' || e[n] == '=')) return e[n] == '=' ? e + n + 1 : "";
        e = strchr(e, ',');
        if (e) e++;
    }
    return NULL;
}

typedef struct qgdk_ctx {
    int N, Np, c, cp, n_ops, n_ess, m, nt, n_pcof, nc_max;
    int have_guard, have_target, inv_batch;
    int cost_type;      // 0 :Infidelity, 1 :Tracking, 2 :Norm (eval_grad_discrete_adjoint.jl:26-35); the kernels get have_target * (1 + cost_type)
    double dt, tf;
    hipStream_t stream;
    // device buffers
    double *ops;        // [(2+2 n_ops)][Np*Np] column-major planes: K_sys, S_sys, Asym_1, Sym_1, ...
    // sparse-operator path (qgd_k_sparse.hip): ELL over the union pattern of all operators, and
    // one ELL list per control operator.  Padding entries point at the row itself with value 0.
    int use_sparse, ell_z, op_z;
    int fuse_terminal;  // set for one call of qgdk_adjoint_blocks: its first launch carries the terminal condition (k_terminal's work)
    int32_t *ell_col;   // [ell_z][Np]
    uint8_t *ell_inv;   // [Np][Np]: slot of column c in row r of the union pattern, 0xff when absent
    double *ell_val;    // [(2+2 n_ops)][ell_z][Np]   planes in the order of `ops`
    int32_t *op_col;    // [n_ops][op_z][Np]
    double *op_val;     // [n_ops][2][op_z][Np]       (Asym_o, Sym_o)
    double *guard;      // [2N][2N] column-major
    double *guard_diag; // [2N] when the projector is diagonal (have_guard == 2)
    double *target;     // panel [Np][2cp]
    double *G;          // control basis, per control [pq][nt][m+1][ncoef_k]
    int64_t *goff;      // offset of control k in G
    int32_t *ncoef;     // coefficients of control k
    int32_t *poff;      // offset of control k in pcof
    double *tab;        // [nt][m+1][n_ops][2]
    double *D;          // [nt][m][Np][2Np]
    double *L, *R;      // [nt][Np][2Np]
    double *LinvA, *LinvT; // [nt][2][Np*Np]
    double *Pr;         // [nt-1][Np][2Np]
    double *Pc;         // [nt-1][2][Np*Np]
    double *hist;       // [nt][Np][2cp]   psi_0 (the state) at every time point
    double *dpsi;       // [nt][m][Np][2cp] derivatives 1..m
    double *forcing;    // [nt][Np][2cp]
    double *yhist;      // [nt][Np][2cp]
    double *lam;        // [nt][Np][2cp]
    double *sigma;      // [sigma_planes][nt][n_ops][m][2]: one plane per column group for the kernels whose grid is
                        // (column group, time point) -- each workgroup STORES its plane, k_contract adds the planes in
                        // order (no atomics: the gradient is bitwise reproducible); the other kernels add into plane 0
    int sigma_planes;   // planes allocated (cp / 8; N > 64: one per contributing tile, qgdk_dense_sigma_planes_max)
    double *cpart;      // [time chunks of k_contract][n_pcof] partial sums, added in chunk order by k_contract_sum
    double *grad;       // [n_pcof]
    double *scal;       // [4]: <w,R>, <w,T>, guard, spare
    double *gpart;      // per-workgroup partial guard penalties, added in index order by the terminal stage (gpart_on: single GPU, resident grid)
    int gpart_on, gpart_n, gpart_terminal;   // gpart_terminal: the terminal stage of this handle adds the partials up (else qgdk_guard_fold does)
    double *term_part;  // [2 * 1024 + 1]: per-workgroup partial overlaps of k_terminal_sum and its ticket counter (zeroed at creation)
    double *cw;         // [2*(m+1)]: c_j dt^j, c_j (-dt)^j
    double *binv;       // work space of the block Gauss-Jordan inverse for N > 64 (qgdk_dense_inverse), or null
    double *inv_scratch;
    double *panel_scratch; // large N: per-workgroup panel slabs of k_derivs/k_gradsweep in HBM instead of LDS
    // large N, GEMM-style kernels (qgd_k_dense.hip): A_d(t_n), D_j(t_n) and {S_o, K_o} in MFMA fragment order
    int dense_gemm;
    double *Afrag, *Dfrag, *OpFrag;
    double *Tlam;       // [nt][2][Np][2Np]: Lambda+ = lambda_{n+1} psi_0^H, Lambda- = lambda_n psi_0^H (third form of the gradient scalars), or null
    double *Xfrag;      // [nt][m] the same matrices as Xouter in fragment order {Re, Im} (left operand of k_ginner_d's 3M tiles), or null
    double *Xouter;     // [nt][m][Np][2Np]: X_j = (1/j) g_j psi_0^H of the gradient scalars' outer-product form (k_gouter), or null
    // blocked scan + time partition (DESIGN.md "Multi-GPU").  The handle covers the time points
    // [n_off, n_off + nt) of a global grid of nt_glob points; blocks [blk_lo, blk_hi) of scan_blocks.
    double *PiX;        // own blocks: [B x PiC | B x PiR], 2*Np*Np doubles each (B = scan_blocks = blocks of this rank)
    double *phiX;       // own blocks: [B x phi], Np*2cp doubles each
    double *bnd, *bndY; // [B+1][Np][2cp] states at the boundaries of the own blocks
    double *RX;         // exchange buffer 0: per rank [R planes | R panel], R = product of the rank's window
    double *phiRX;      // exchange buffer 1: per rank [phi^rank | y_N (last rank)]
    double *wbnd, *wbndY; // [W+1][Np][2cp] scratch of the chains over the windows
    double *psi0;       // initial panel [Np][2cp]
    double *zero_panel; // [Np][2cp] of zeros
    double *redbuf;     // [n_pcof + 4]: grad followed by scal (one all-reduce)
    int scan_blocks, scan_blen, bpr, blk_lo, blk_hi, blocks_glob;   // scan_blocks = bpr (blocks of this rank); blk_lo/hi: its global block range
    // second scan level over the block propagators: scan_blocks2 super-blocks of scan_g blocks
    int scan_blocks2, scan_g;
    double *PiC2, *PiR2, *phi2, *bnd2, *bndY2;
    // forward history pass over sub-blocks (N <= 64, one rank): stored running products of the block chains (Hmid:
    // [B][sub_n] after 3, 6, ... steps) and of the level-2 chains (Qmid: [B2][g-2] after 2 .. g-1 blocks)
    int sub_hist, sub_n;
    double *Hmid, *Qmid;
    // adjoint history pass: suffix products of the blocks of every super-block (panel layout, [B2][g-2]) and their affine
    // parts ([B2][g-2] panels) -- the block-level prefix of the pass is then one step (qgd_k_chain.hip, ChainArgs::suf_P)
    double *SufP, *SufPhi;
    int part_rank, part_world, n_off, nt_glob;
    // chunked time grid (bounded memory, qgd_set_memory_budget): the handle's per-time-point buffers hold ONE window of
    // the grid at a time.  The control basis G stays whole: g_nt time points, the current window starts at g_n0
    // (g_nt = 0: the basis covers exactly the nt points of the buffers).  keep_scal: k_tables must not clear the
    // scalars (a later window of the same evaluation); grad_accumulate: k_contract adds to grad instead of storing it.
    int g_nt, g_n0, keep_scal, grad_accumulate;
    // forced (forward-sensitivity) gradient, qgd_k_forced.hip: basis responses and sensitivity scan buffers
    double *fs_BR, *fs_BL;   // [nt][n_ops*2*m][Np][2cp]
    double *fs_phi, *fs_bnd; // [B][Np][2cpS], [B+1][Np][2cpS], cpS = n_pcof * cp
    double *fs_gacc;         // [n_pcof]
    double *fs_scratch;      // per-workgroup panel slabs of k_forced_basis / k_forcing_terms when they exceed the LDS (N > 64), else null
    // eval_forward with a user forcing: F, E [nt][m][Np][2cp]; XR, XL, Q [nt][Np][2cp]; scan buffers
    double *ff_F, *ff_E, *ff_XR, *ff_XL, *ff_Q, *ff_phi, *ff_bnd;
    int *status;
    double cw_host[2 * 20];
    int32_t ncoef_host[QGD_MAX_OPS_DEV], poff_host[QGD_MAX_OPS_DEV];      // host copies of ncoef / poff / goff (kernel arguments of qgd_k_tiny.hip)
    int64_t goff_host[QGD_MAX_OPS_DEV];
    // Result mirror (single GPU, resident grid): the last kernel of a gradient evaluation, k_contract_sum, also writes
    // [grad | scal(4) | status] into pinned host memory and then a sequence number the host is polling -- no copy packet
    // behind the kernel and no wait for the stream's completion signal (qgd_host_eval.cpp: fetch_results).  mirror_dev: the
    // device-visible address of that host buffer ([n_pcof + 6] doubles, the sequence number in the last one as a 64-bit
    // integer), or null for this launch; mirror_ticket: a zeroed device counter.
    double *mirror_dev;
    unsigned long long mirror_seq;
    unsigned int *mirror_ticket;
    // Fused front (qgd_front.h; Np = 64, sparse operators, one rank, resident grid): front = 1 for the evaluation in flight --
    // the step propagators are the same-point products S_n = R_n L_n^-1 (Pc / Pr hold S_n, LinvT holds L_n^-H, L / R hold
    // L_n^H / R_n^H), the forward scan runs in phi_n = L_n psi_n from phi0 into phist, k_psi turns it into the state history,
    // the guard forcing f_n and h_n = L_n^-H f_n (hforc), and the adjoint scan runs directly in lambda with the forcing h.
    int front;
    double *phi0;       // [Np][2cp]   L_0 psi_0
    double *phist;      // [nt][Np][2cp] (the buffer of yhist: the front path has no y)
    double *hforc;      // [nt][Np][2cp]
    double *termU;      // [Np][2cp]   L_N^-H target: lambda_N = (2/N_ess^2)(a + ib) termU + h_N
} qgdk_ctx;

#ifdef __cplusplus
extern "C" {
#endif
int qgdk_tables(const qgdk_ctx *c, const double *pcof_dev);
int qgdk_tables_from_host(const qgdk_ctx *c, const double *pt_dev, const double *qt_dev);
#define QGD_PCOF_KERNARG 448      /* doubles of pcof that fit beside the other kernel arguments (4 KB) */
int qgdk_tables_kernarg(const qgdk_ctx *c, const double *pcof_host, int n_pcof);
int qgdk_front_pre_plan(const qgdk_ctx *c, int *q2, int *q1);   /* how many second / first workgroups of k_front start from pre-built step matrices; returns extra */
int qgdk_front_supported(const qgdk_ctx *c);   /* Np = 64, sparse operators, order <= 8, the build's LDS beside the elimination's */
int qgdk_tables_front(const qgdk_ctx *c, const double *pcof_host, int n_pcof);   /* tables + the pre-built step matrices + phi_0 */
int qgdk_front(const qgdk_ctx *c);             /* L_n^-H, S_n for every time point: build + elimination in one workgroup */
int qgdk_psi(const qgdk_ctx *c);               /* psi_n = L_n^-1 phi_n, guard forcing and penalty, h_n = L_n^-H f_n, termU */
int qgdk_build_LR(const qgdk_ctx *c);
int qgdk_inverse(const qgdk_ctx *c);
int qgdk_propagator(const qgdk_ctx *c);
int qgdk_propagator_is_fused(const qgdk_ctx *c);
int qgdk_forward_blocks(const qgdk_ctx *c);
int qgdk_forward_finish(const qgdk_ctx *c);
int qgdk_forward_blocks_range(const qgdk_ctx *c, int b0, int b1, hipStream_t stream);
int qgdk_forward_blocks_upper(const qgdk_ctx *c);
int qgdk_guard(const qgdk_ctx *c);
int qgdk_guard_is_fused(const qgdk_ctx *c);
int qgdk_guard_parts(const qgdk_ctx *c);
int qgdk_guard_fold(const qgdk_ctx *c);       /* scal[2] += the partials of gpart in a fixed order (windows, ranks without the final time) */      /* workgroups of the guard stage of this configuration = entries of gpart it writes */
int qgdk_terminal_can_fuse(const qgdk_ctx *c);
int qgdk_terminal(const qgdk_ctx *c, int write_y);
int qgdk_terminal_given(const qgdk_ctx *c);   /* y_N from the overlaps already in scal (column shards) */
int qgdk_adjoint_blocks(const qgdk_ctx *c);
int qgdk_adjoint_finish(const qgdk_ctx *c);
int qgdk_apply_LH(const qgdk_ctx *c);
int qgdk_lambda(const qgdk_ctx *c);
int qgdk_derivs(const qgdk_ctx *c);
int qgdk_gradient(const qgdk_ctx *c);
int qgdk_contract(const qgdk_ctx *c);
int qgdk_gradient_needs_derivs(const qgdk_ctx *c);
int qgdk_apply(const qgdk_ctx *c, const double *in_dev, double *out_dev, int n, int d, double sign);
int qgdk_adjoint_derivs(const qgdk_ctx *c, double *dlam, double *scratch);
int qgdk_tiny_supported(const qgdk_ctx *c, int n_pcof);      // small problems (N <= 4): four launches instead of twelve (qgd_k_tiny.hip)
int qgdk_tiny_eval(const qgdk_ctx *c, const double *pcof_host, int n_pcof, int gradient);
int qgdk_contract_rows(const qgdk_ctx *c, int rows);         // cpart rows -> grad (+ host mirror), fixed order
int qgdk_mirror_scalars(const qgdk_ctx *c);      // result mirror of an evaluation without a gradient: [scal | status | sequence number]
size_t qgdk_lds_needed(int Np, int m, int n_ops);
int qgdk_sparse_supported(int Np, int m, int n_ops, int Z);
size_t qgdk_forced_lds(int Np, int m);
int qgdk_forced_basis(const qgdk_ctx *c);
int qgdk_forced_chains(const qgdk_ctx *c);
int qgdk_forcing_terms(const qgdk_ctx *c);
int qgdk_forcing_add_derivs(const qgdk_ctx *c);
int qgdk_forcing_sweep(const qgdk_ctx *c);
int qgdk_guard_kernel(const qgdk_ctx *c);
int qgdk_build_LR_sparse(const qgdk_ctx *c);
int qgdk_dense_operator_frag(const qgdk_ctx *c);
int qgdk_dense_build_LR(const qgdk_ctx *c);
int qgdk_dense_sigma_planes(const qgdk_ctx *c);   // planes of sigma the N > 64 gradient kernels write for the form in use
int qgdk_dense_sigma_planes_max(int Np, int cp, int m);
int qgdk_dense_sigma_form(const qgdk_ctx *c);     // 0..3: which form of the gradient scalars the N > 64 path takes (qgd_k_dense.hip)
int qgdk_dense_chain_step(hipStream_t stream, int adj, const double *P, const double *in, double *out, const double *forcing, int Np, int cp);
int qgdk_dense_inverse(const qgdk_ctx *c);        // 1: the block Gauss-Jordan inverse took the launch, 0: not taken
size_t qgdk_dense_inverse_words(int Np, int nt);  // doubles of work space it needs
int qgdk_inverse_diag(const qgdk_ctx *c, const double *Win, size_t mstride, int ldw, size_t off, int bs, double *DkC, int *flags);
int qgdk_inverse_redo(const qgdk_ctx *c, const int *flags);
int qgdk_dense_propagator(const qgdk_ctx *c);     // 1: launched P = Linv R on the three-product tiles, 0: not taken
int qgdk_dense_derivs(const qgdk_ctx *c);
int qgdk_dense_gradient(const qgdk_ctx *c);
int qgdk_dense_gradient_needs_derivs(const qgdk_ctx *c);
int qgdk_dense_lambda(const qgdk_ctx *c);
int qgdk_gradient_sparse(const qgdk_ctx *c);
int qgdk_derivs_sparse(const qgdk_ctx *c);
/* qgd_k_layout.hip: panels [n][j][Np][2cp] -> reference layout dst[col][n][j][2N] (to_panels = 0) or back */
int qgdk_flag_to_scal(const qgdk_ctx *c);   /* scal[3] = 1.0 when the singularity flag is set (travels in the all-reduced range) */
int qgdk_layout(const qgdk_ctx *c, const double *panels, long long src_n, long long src_j, double *ref,
                long long dst_col, long long dst_n, long long dst_j, int n0, int n_cnt, int j_cnt,
                int to_panels, hipStream_t stream, int max_workgroups);
#ifdef __cplusplus
}
#endif
#endif
