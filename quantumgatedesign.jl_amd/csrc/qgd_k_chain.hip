// qgd_k_chain.hip -- the two sweeps as blocked scans, guard, terminal condition, lambda
// (conventions and layouts: qgd_kernels_common.h; algorithm: DESIGN.md)
#include "qgd_kernels_common.h"

// ---------------------------------------------------------------------------
// K4/K7: the two sweeps as a blocked scan over time.
//   forward: psi_{n+1} = P_n psi_n                       (forward_evolution.jl:163-221)
//   adjoint: y_n = P_n^H y_{n+1} + f_n                   (forward_evolution.jl:421-462 in the
//            variable y_n = L_n^T lambda_n)
// The S = nt-1 steps are cut into B blocks of `blen` steps.  Three phases:
//   (i)   per block, in parallel: the block propagator Pi_b = P_{e-1}...P_s (forward; its
//         conjugate transpose serves the adjoint) by chaining the identity's columns, and
//         for the adjoint the affine part phi_b (zero start, forcing added);
//   (ii)  one short sequential chain over the B block propagators -> states at block starts;
//   (iii) per block, in parallel: re-run the block from its true start, writing the history.
// Chain length drops from S to 2*blen + B matrix-panel products.
// All three phases are the same "chain" kernel; one workgroup = one (block, group of 8
// columns).  MODE 0: identity start, store Pi_b.  MODE 1: forward, write history.
// MODE 2: adjoint from zero, store phi_b.  MODE 3: adjoint, write history.
// ---------------------------------------------------------------------------
struct ChainArgs {
    const double *Pmat;      // forward: planes [S][2][Np*Np] col-major; adjoint: panels [S][Np][2Np]
    const double *start;     // start panels, block b at start + b*start_stride  (layout [Np][2cp])
    long long start_stride;
    double *out;             // history [.][Np][2cp] (MODE 1: out[n+1], MODE 3: out[n])
    const double *forcing;   // [.][Np][2cp], adjoint modes
    double *PiC, *PiR;       // MODE 0 outputs: planes / panel per block
    double *phi;             // MODE 2 output: [B][Np][2cp]
    int Np, cp, S, nblocks, blen, ngroups;
    // exchange-buffer addressing (multi-GPU layout, one chunk per rank):
    // matrix n lives at Pmat + (n / pm_bpr) * pm_chunk + (n % pm_bpr) * 2*Np*Np   (pm_bpr = 0: n * 2*Np*Np)
    int pm_bpr; long long pm_chunk;
    // forcing/history slot of time index n is n + n / f_bpr                          (f_bpr = 0: n)
    int f_bpr;
    // MODE 1 with a diagonal guard projector (multi_qudit_systems.jl:316-349): the history pass also
    // writes the adjoint forcing f_n = -(2 dt/tf) trap_n W w_n and accumulates the guard penalty
    // (dt/tf) sum_n trap_n w_n^T W w_n (infidelity.jl:56-96) of the states it produces
    const double *guard_diag;   // [2N] or null
    double *gpart;              // MODE 1 with the guard fused: partial penalty of workgroup bid (stored; null: atomicAdd into scal[2])
    const double *t_gpart; int t_gpart_n;      // t_on: the partial penalties the terminal workgroup adds up in index order
    double *guard_forcing;      // [.][Np][2cp]
    double *scal;               // scal[2] += penalty
    int gN, n_off, nt_glob, count_first;
    double dt, tf;
    // MODE 1/3 prefix: before its own steps the workgroup of block b advances the common start state over
    // the coarser levels of the scan -- windows of the other ranks (kind 0), super-blocks (kind 1), blocks
    // (kind 2) -- so ONE launch replaces the chains over the levels (the prefixes are redundant work on
    // otherwise idle CUs).  Forward: kind 0 n = 0..cnt-1, kind 1 n = 0..j-1, kind 2 n = first..b-1 with
    // j = b / pre_g; adjoint: the mirror image from the top.
    int npre, pre_kind[3], pre_g, pre_B2, pre_rank_count;
    const double *pre_P[3], *pre_f[3];
    int pre_pm_bpr[3], pre_f_bpr[3]; long long pre_pm_chunk[3];
    double *pre_start_out;   // block 0 (forward) / the last block (adjoint) stores the state after the kind-0 segment here
    // MODE 4/5 (forward sensitivities of the forced gradient, eval_grad_forced.jl:17-194): one column
    // group per (control parameter, state column group).  fs_mode 1: the forcing of step n is
    // assembled from the 2m basis responses of the parameter's control,
    //   q_n = sum_{tau,d} G^tau_l(n,d) BR[n][b] - G^tau_l(n+1,d) BL[n+1][b],  b = (k*2+tau)*m + d
    int fs_mode, fs_m, fs_nops, fs_gpc, fs_nt;   // gpc: state column groups per parameter
    int fs_n0;                                  // global index of the local time point 0 in the control basis (a window of a long grid)
    const double *fs_BR, *fs_BL;                 // [nt][NB][Np][2*cp_state], already multiplied by L^-1
    const double *fs_G; const int64_t *fs_goff; const int32_t *fs_ncoef, *fs_poff;
    const double *fs_gf;                         // adjoint forcing f_n [nt][Np][2*cp_state]: guard part = -<f_n, s_n>
    double *fs_gacc;                             // [n_pcof] guard sums
    // MODE 0, mid_every > 0: the running product is also stored after k = mid_first, mid_first + mid_every, ... (< block
    // length) matrices, planes layout, at mid_out + ((size_t)b * mid_n + (k - mid_first) / mid_every) * 2 Np^2.
    // MODE 1, sub_T > 0 (forward history pass over SUB-blocks): `nblocks` counts sub-blocks, sub_T per block of blen steps,
    // sub_len steps each; sub-block t >= 1 of block b starts from the block's start state advanced by ONE step with the
    // stored product sub_H[b * sub_n + t - 1] (the first t * sub_len step matrices of the block).  pre_Q: the prefix
    // products of the blocks inside a super-block stored by the level-2 scan, pre_Q[j * pre_Qn + i - 2] = product of the
    // first i >= 2 blocks of super-block j: the block-level prefix of the history pass is then ONE step instead of i.
    // Together: <= B2 + 1 + 1 + sub_len dependent steps instead of B2 + g + blen, on sub_T times as many CUs.
    double *mid_out; int mid_every, mid_n, mid_first;
    int sub_T, sub_len, sub_n, pre_Qn;
    const double *sub_H, *pre_Q;
    // Adjoint history pass (MODE 3) with the block-level prefix in ONE step: the level-2 launches also store
    //   suf_P[j * suf_n + cnt - 2]   = Pi_{e-1} ... Pi_{e-cnt} in panel layout (MODE 6 chains beside the level-2 products),
    //   suf_phi[j * suf_n + cnt - 2] = the affine part of those cnt blocks (running values of the MODE 2 level-2 chain),
    // cnt = 2 .. g-1, e = end of super-block j: y at the end of block bb is then suf_P^H y_e + suf_phi instead of cnt
    // steps (cnt = 1 is the block propagator / affine part itself).  MODE 2: mid_out/mid_n describe the affine stores.
    const double *suf_P, *suf_phi; int suf_n;
    // MODE 2, t_on: ONE extra workgroup (the last of the grid) does the work of k_terminal -- overlaps <w_N,R>, <w_N,T>
    // and y_N -- beside the affine parts of the blocks, which do not need it: the adjoint sweep starts one launch earlier
    int t_on, t_nt, t_ness, t_have_target;
    const double *t_hist, *t_target, *t_forcing, *t_ytarget;
    // MODE 0, p0_on (fused front, qgd_front.h): ONE extra workgroup (the last of the grid) forms phi_0 = L_0 psi_0 from the panel
    // of L_0^H beside the block products, which do not need it (the history pass, two launches later, starts from it)
    int p0_on;
    const double *p0_E, *p0_psi0;
    double *p0_out;
    double *t_yhist, *t_scal, *t_y2, *t_y3, *t_y4;
};

__device__ __forceinline__ void terminal_block(const double *__restrict__ hist, const double *__restrict__ target,
                                               const double *__restrict__ forcing, double *__restrict__ yhist,
                                               double *__restrict__ scal, int Np, int cp, int nt, int n_ess, int have_target,
                                               int write_y, double *__restrict__ y2, double *__restrict__ y3,
                                               double *__restrict__ y4, int given_ab, const double *gpart = nullptr, int gpart_n = 0,
                                               const double *ytarget = nullptr);

// phi_0 = L_0 psi_0 = E^H psi_0 (E = L_0^H, panel layout) as one MFMA product: the left operand is read the way the adjoint
// sweep reads a step matrix (P^H from the panel of P), all 32 fragment loads of a wave in flight at once; waves 0-3 of the
// workgroup take the four row blocks.  (A loop over k with one thread per output element took 15 us inside a 25 us launch.)
__device__ __forceinline__ void phi0_block(const double *__restrict__ E, const double *__restrict__ psi0, double *__restrict__ phi0, const int cp)
{
    __shared__ double ps[64 * 16];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, c16 = lane & 15, kk = lane >> 4, PWc = 2 * cp;
    const int arow = (wave & 3) * 16 + c16;
    double are[16], aim[16];
    #pragma unroll
    for (int i = 0; i < 16; i++) {      // (E^H)(row, k) = conj(E(k, row))
        const double *P = E + (size_t)(4 * i + kk) * 128 + (arow >> 3) * 16 + (arow & 7);
        are[i] = P[0]; aim[i] = P[8];
    }
    for (int g = 0; g < cp / 8; g++) {
        __syncthreads();
        for (int e = threadIdx.x; e < 64 * 16; e += blockDim.x) ps[e] = psi0[(size_t)(e >> 4) * PWc + 16 * g + (e & 15)];
        __syncthreads();
        if (wave < 4) {
            d4 acc0 = (d4){0, 0, 0, 0}, acc1 = (d4){0, 0, 0, 0};
            #pragma unroll
            for (int i = 0; i < 16; i++) {
                double b1, b2;
                panel_b(ps + (size_t)(4 * i + kk) * 16, c16, b1, b2);
                acc0 = MFMA(are[i], b1, acc0);
                acc1 = MFMA(-aim[i], b2, acc1);
            }
            #pragma unroll
            for (int r = 0; r < 4; r++) phi0[(size_t)(wave * 16 + kk + 4 * r) * PWc + 16 * g + c16] = acc0[r] + acc1[r];
        }
    }
}

__host__ __device__ __forceinline__ const double *chain_matrix(const ChainArgs &a, int n)
{
    const size_t pl2 = (size_t)2 * a.Np * a.Np;
    if (a.pm_bpr) return a.Pmat + (size_t)(n / a.pm_bpr) * a.pm_chunk + (size_t)(n % a.pm_bpr) * pl2;
    return a.Pmat + (size_t)n * pl2;
}

template <int MODE>
__device__ __forceinline__ void chain_block_of(const ChainArgs &a, int &b, int &grp, int bid = -1)
{
    if (bid < 0) bid = blockIdx.x;
    if (MODE == 0 || MODE == 6) {
        // XCD-aware: the ngroups workgroups of one block share blockIdx%8, hence one XCD's L2,
        // because they all stream the same P_n (speed only; MI355X_MICROARCH.md "Workgroup dispatch").
        const int xcd = bid & 7, slot = bid >> 3;
        b = xcd + 8 * (slot / a.ngroups);
        grp = slot % a.ngroups;
    } else {
        b = bid / a.ngroups;
        grp = bid % a.ngroups;
    }
}

// A fragment of step n at (row, k)
template <bool ADJ>
__device__ __forceinline__ void chain_a(const double *__restrict__ Pn, int Np, int arow, int k,
                                        double &are, double &aim)
{   // Pn: base of this step's matrix
    const size_t pl = (size_t)Np * Np;

    if (!ADJ) {
        const double *P = Pn + (size_t)arow + (size_t)Np * k;
        are = P[0];
        aim = P[pl];
    } else {   // (P^H)(row,k) = conj(P(k,row)); P stored as panel
        const double *P = Pn + (size_t)k * 2 * Np + (arow >> 3) * 16 + (arow & 7);
        are = P[0];
        aim = -P[8];
    }
}

// same, but the adjoint's imaginary part is returned un-negated (sign = -1 is applied to the B
// operand instead): nothing touches the loaded registers until the MFMA, so the loads stay in flight
template <bool ADJ>
__device__ __forceinline__ void chain_a_raw(const double *__restrict__ Pn, int Np, int arow, int k,
                                            double &are, double &aim)
{
    const size_t pl = (size_t)Np * Np;
    if (!ADJ) {
        const double *P = Pn + (size_t)arow + (size_t)Np * k;
        are = P[0];
        aim = P[pl];
    } else {
        const double *P = Pn + (size_t)k * 2 * Np + (arow >> 3) * 16 + (arow & 7);
        are = P[0];
        aim = P[8];
    }
}

// NP > 0: compile-time size.  The steps of a chain are sequential, but the LEFT operands (P_n, and the forcing
// of the adjoint) do not depend on the state: NT teams of NP/16 waves take the steps round-robin, team t
// computing steps t, t+NT, ...  Each team issues the loads of its next step right after finishing one and then
// sits out the steps of the other teams at the step barriers, so its operands have NT-1 step times to arrive --
// with plain compiler-managed registers.  One barrier per step, an LDS-only one (see lds_barrier); the state is
// double buffered in LDS; a wave owns 16 rows over the full K range (no partial sums), the history is stored
// straight from the accumulators.  NG column groups per workgroup share the A fragments.
// Teams (scripts/ubench/chain_bench.hip, 64 blocks, us per launch; mode 0 with NG=2):
//                              mode 0, 9 steps   mode 1, 24   mode 2, 9   mode 3, 24
//   4 teams, __syncthreads     30.5              ~53          18.5        ~50        (128-VGPR cap: modes 1, 3 spill)
//   3 teams, LDS-only barrier,
//            buffer addressing 32.7              37.1         17.3        39.0
//   2 teams, the same          28.7              33.7         14.9        34.1
// QGD_CHAIN_NT (compile time) overrides the choice.
#ifdef QGD_CHAIN_NT
#define CHAIN_NT(MODE) QGD_CHAIN_NT
#else
#define CHAIN_NT(MODE) 2
#endif
#ifdef QGD_CHAIN_PROFILE
__device__ long long g_chain_prof[64 * 8];
#endif
// MODE 6 (round 3): the adjoint matrix chain X <- Pi_b^H X from the identity over the blocks of a super-block, last block
// first, with P^H read from the PANEL copies of the block propagators; its running values are the conjugate transposes of
// the suffix products Pi_{e-1} ... Pi_{e-cnt}, stored in panel layout for the adjoint history pass (ChainArgs::suf_P).
template <int NP, int MODE, int NG>
__device__ __forceinline__ void chain_fast_body(const ChainArgs &a, const int bid, const int nbid)
{
    constexpr bool ADJ = (MODE == 2 || MODE == 3 || MODE == 6);      // backward in time with P^H
    constexpr bool FORC = (MODE >= 2 && MODE != 6);       // affine: a forcing term is added every step
    constexpr bool ZERO = (MODE == 2 || MODE == 4);       // zero start, the final state is the block's affine part
    constexpr bool MAT = (MODE == 0 || MODE == 6);        // the state is a whole matrix (NP columns), not a panel of the problem's columns
    constexpr int NRB = NP / 16, NT = CHAIN_NT(MODE), KST = NP / 4, NTH = NP * 4 * NT;
    __shared__ __attribute__((aligned(16))) double part[2][NG][NP * 16];

    if (MODE == 0 && a.p0_on && bid == nbid - 1) { phi0_block(a.p0_E, a.p0_psi0, a.p0_out, a.cp); return; }
    if (MODE == 2 && a.t_on && bid == nbid - 1) {      // the extra workgroup: k_terminal's work
        terminal_block(a.t_hist, a.t_target, a.t_forcing, a.t_yhist, a.t_scal, NP, a.cp, a.t_nt, a.t_ness, a.t_have_target, 1,
                       a.t_y2, a.t_y3, a.t_y4, 0, a.t_gpart, a.t_gpart_n, a.t_ytarget);
        return;
    }
    int b, grp0;
    {
        ChainArgs a2 = a; a2.ngroups = a.ngroups / NG;
        chain_block_of<MODE>(a2, b, grp0, bid);
        grp0 *= NG;
    }
    if (b >= a.nblocks) return;
    const bool subs = (MODE == 1) && a.sub_T > 0;        // b counts sub-blocks: block bb, sub-block tsub
    const int bb = subs ? b / a.sub_T : b, tsub = subs ? b % a.sub_T : 0;
    const int blk_end = ((bb + 1) * a.blen < a.S) ? (bb + 1) * a.blen : a.S;
    const int s0 = bb * a.blen + (subs ? tsub * a.sub_len : 0);
    const int e0 = subs ? ((s0 + a.sub_len < blk_end) ? s0 + a.sub_len : blk_end) : blk_end;
    if (subs && s0 >= blk_end) {                         // (a short last block has fewer sub-blocks)
        if (MODE == 1 && a.guard_diag && a.gpart && threadIdx.x == 0) a.gpart[bid] = 0.0;      // ... whose guard partials are zero, not stale
        return;
    }
    const int PWc = MAT ? 2 * NP : 2 * a.cp;
    const size_t hstep = (size_t)NP * PWc;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int c16 = lane & 15, kk = lane >> 4;
    const int team = wave / NRB, rb = wave % NRB;
    const int arow = rb * 16 + c16;
    // left-operand fragment of step matrix P at (arow, 4 i + kk): planes (forward) / panel read as P^H (adjoint, the
    // imaginary part un-negated: the sign goes to the B operand)
    constexpr int A_STEP = ADJ ? 8 * NP : 4 * NP, A_IM = ADJ ? 8 : NP * NP;
    const int a_lane = 8 * (ADJ ? kk * 2 * NP + (arow >> 3) * 16 + (arow & 7) : arow + NP * kk);     // bytes
    const int o_lane = 8 * ((rb * 16 + kk) * PWc + c16);                                                // bytes        // this lane's element (row rb*16+kk, column c16) of a state panel
    const int nsteps = (e0 > s0) ? e0 - s0 : 0;
    double pen = 0.0;                                     // guard penalty of the states this thread handles
    // prefix segments (MODE 1/3): counts and first indices for this block
    int pcnt[4] = {0, 0, 0, 0}, pfirst[4] = {0, 0, 0, 0}, npfx = 0;
    const double *pfix[4] = {nullptr, nullptr, nullptr, nullptr};   // a segment of ONE step with this stored product
    const double *pffix[4] = {nullptr, nullptr, nullptr, nullptr};  // ... and (adjoint) this stored affine part
    const size_t pl2_ = (size_t)2 * NP * NP;
    if ((MODE == 1 || MODE == 3) && a.npre > 0) {
        bool have1 = false;
        for (int q = 0; q < a.npre; q++) have1 = have1 || (a.pre_kind[q] == 1);
        const int nblk = subs ? a.nblocks / a.sub_T : a.nblocks;
        const int j = have1 ? bb / a.pre_g : 0;
        const int ej = have1 ? (((j + 1) * a.pre_g < nblk) ? (j + 1) * a.pre_g : nblk) : nblk;
        for (int q = 0; q < a.npre; q++) {
            const int kind = a.pre_kind[q];
            if (!ADJ) {
                pfirst[q] = (kind == 2 && have1) ? j * a.pre_g : 0;
                pcnt[q] = (kind == 0) ? a.pre_rank_count : (kind == 1) ? j : bb - pfirst[q];
                if (MODE == 1 && kind == 2 && a.pre_Q && have1 && pcnt[q] >= 1) {   // the stored block-prefix product
                    const int i = pcnt[q];
                    pfix[q] = (i == 1) ? a.pre_P[q] + (size_t)pfirst[q] * pl2_ : a.pre_Q + ((size_t)j * a.pre_Qn + (i - 2)) * pl2_;
                    pcnt[q] = 1;
                }
            } else {    // descending: first = highest index
                pfirst[q] = (kind == 0) ? a.pre_rank_count - 1 : (kind == 1) ? a.pre_B2 - 1 : ej - 1;
                pcnt[q] = (kind == 0) ? a.pre_rank_count : (kind == 1) ? a.pre_B2 - 1 - j : ej - 1 - bb;
                if (MODE == 3 && kind == 2 && a.suf_P && have1 && pcnt[q] >= 2) {   // the stored suffix product and affine part of those blocks
                    const int cnt = pcnt[q];
                    pfix[q] = a.suf_P + ((size_t)j * a.suf_n + (cnt - 2)) * pl2_;
                    pffix[q] = a.suf_phi + ((size_t)j * a.suf_n + (cnt - 2)) * hstep;
                    pcnt[q] = 1;
                }
            }
            if (pcnt[q] < 0) pcnt[q] = 0;
            npfx += pcnt[q];
        }
    }
    int npre_eff = ((MODE == 1 || MODE == 3) ? a.npre : 0);
    if (subs && tsub > 0) {                               // from the block's start state to this sub-block's: one step
        pfix[npre_eff] = a.sub_H + ((size_t)bb * a.sub_n + (tsub - 1)) * pl2_;
        pcnt[npre_eff] = 1; npfx += 1; npre_eff++;
    }
    const int total = npfx + nsteps;

    // start state into part[0]
    auto step_index = [&](int st) { return ADJ ? e0 - 1 - st : s0 + st; };
    // The first step of a chain that starts from the identity (MODE 0) or from zero (MODE 2) needs no product: its
    // result is the step matrix itself / the forcing.  It is loaded straight into the buffer step 1 reads.
    constexpr bool SKIP0 = (MODE == 0 || MODE == 2 || MODE == 6);
    const int first = (SKIP0 && total > 0) ? 1 : 0;
    double are[KST], aim[KST], fo[NG][4];
    const double sgn2 = ((c16 < 8) != ADJ) ? -1.0 : 1.0;  // sign of the Aim [Bim|Bre] sum in this lane's output column
    double gw[4] = {0.0, 0.0, 0.0, 0.0};                 // guard weights of this lane's accumulator elements
    if (MODE == 1 && a.guard_diag) {
        #pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = rb * 16 + kk + 4 * r;
            gw[r] = (row < a.gN) ? a.guard_diag[row + ((c16 >= 8) ? a.gN : 0)] : 0.0;
        }
    }
    auto issue = [&](int st) {                            // left operand (and forcing) of step st
        int n, fbpr = a.f_bpr;
        const double *Pn, *fsrc = a.forcing;
        // (everything here is wave-uniform; readfirstlane tells the compiler so, and the pointers stay in SGPRs)
        if (st >= npfx) { n = __builtin_amdgcn_readfirstlane(step_index(st - npfx)); Pn = chain_matrix(a, n); }
        else {                                            // a prefix step
            int q = 0, sl = st;
            while (sl >= pcnt[q]) { sl -= pcnt[q]; q++; }
            q = __builtin_amdgcn_readfirstlane(q);
            n = __builtin_amdgcn_readfirstlane(ADJ ? pfirst[q] - sl : pfirst[q] + sl);
            const size_t pl2 = (size_t)2 * NP * NP;
            if (pfix[q]) Pn = pfix[q];
            else Pn = a.pre_pm_bpr[q] ? a.pre_P[q] + (size_t)(n / a.pre_pm_bpr[q]) * a.pre_pm_chunk[q] + (size_t)(n % a.pre_pm_bpr[q]) * pl2
                                      : a.pre_P[q] + (size_t)n * pl2;
            if (q < a.npre) { fsrc = a.pre_f[q]; fbpr = a.pre_f_bpr[q]; }
            if (pffix[q]) { fsrc = pffix[q]; fbpr = 0; n = 0; }
        }
        // Addresses = buffer descriptor on the step's matrix (SGPRs) + a per-lane byte offset computed once + constants:
        // no vector ALU per load.  While the other team runs its burst of f64 MFMAs, a wave on the same SIMD gets about
        // one VALU instruction per MFMA (64 cycles): with 64-bit addresses built in VGPRs, issuing these 32 loads took
        // 2500 cycles (chain_bench -DQGD_CHAIN_PROFILE) -- longer than the step they are supposed to hide behind.
        const __amdgpu_buffer_rsrc_t rP = buffer_of(Pn);
        #pragma unroll
        for (int i = 0; i < KST; i++) {
            are[i] = buffer_load_f64(rP, a_lane, i * A_STEP * 8);
            aim[i] = buffer_load_f64(rP, a_lane, (i * A_STEP + A_IM) * 8);
        }
        if (FORC && !(MODE >= 4 && a.fs_mode == 1)) {
            const __amdgpu_buffer_rsrc_t rF = buffer_of(fsrc + (size_t)(fbpr ? n + n / fbpr : n) * hstep + (size_t)grp0 * 16);
            #pragma unroll
            for (int g = 0; g < NG; g++)
                #pragma unroll
                for (int r = 0; r < 4; r++)
                    fo[g][r] = buffer_load_f64(rF, o_lane, (4 * r * PWc + g * 16) * 8);
        }
        if (MODE >= 4 && a.fs_mode == 1) {               // assemble the sensitivity forcing of step n -> n+1
            #pragma unroll
            for (int g = 0; g < NG; g++) {
                const int par = (grp0 + g) / a.fs_gpc, cg = (grp0 + g) % a.fs_gpc;
                int k = 0;
                while (k + 1 < a.fs_nops && par >= a.fs_poff[k + 1]) k++;
                const int l = par - a.fs_poff[k], nc = a.fs_ncoef[k];
                const int NB = a.fs_nops * 2 * a.fs_m, PWb = 16 * a.fs_gpc;
                const size_t pstep = (size_t)NP * PWb;
                double accf[4] = {0.0, 0.0, 0.0, 0.0};
                for (int tau = 0; tau < 2; tau++)
                    for (int d = 0; d < a.fs_m; d++) {
                        const double *gk = a.fs_G + a.fs_goff[k];
                        const double g0 = gk[(((size_t)tau * a.fs_nt + n + a.fs_n0) * (a.fs_m + 1) + d) * nc + l];
                        const double g1 = gk[(((size_t)tau * a.fs_nt + n + a.fs_n0 + 1) * (a.fs_m + 1) + d) * nc + l];
                        const int bidx = (k * 2 + tau) * a.fs_m + d;
                        const double *br = a.fs_BR + ((size_t)n * NB + bidx) * pstep + cg * 16 + c16;
                        const double *bl = a.fs_BL + ((size_t)(n + 1) * NB + bidx) * pstep + cg * 16 + c16;
                        #pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const size_t ro = (size_t)(rb * 16 + kk + 4 * r) * PWb;
                            accf[r] += g0 * br[ro] - g1 * bl[ro];
                        }
                    }
                #pragma unroll
                for (int r = 0; r < 4; r++) fo[g][r] = accf[r];
            }
        }
    };
    const int st0 = (team < first) ? team + NT : team;   // this team's first step
    if (st0 < total) issue(st0);                          // (ahead of the start state: the two do not depend on each other)
    // start state into part[first].  The loads of all iterations are issued before the first LDS write (the rolled loop
    // waited for every global round trip in turn: two per workgroup on top of the operand loads above).
    {
        constexpr int NIT = (NG * NP * 16 + NTH - 1) / NTH;
        double sv[NIT];
        #pragma unroll
        for (int it = 0; it < NIT; it++) {
            const int e = tid + it * NTH;
            const int g = e / (NP * 16), el = e % (NP * 16), row = el >> 4, c = el & 15;
            double v = 0.0;
            if (e < NG * NP * 16) {
                if (SKIP0 && first) {
                    const int n0 = step_index(0);
                    if (MODE == 0) v = chain_matrix(a, n0)[(c >= 8 ? (size_t)NP * NP : 0) + row + (size_t)NP * ((grp0 + g) * 8 + (c & 7))];
                    else if (MODE == 6) {   // (Pi^H)[row][col] = conj(Pi[col][row]) from the panel copy of Pi
                        const double pv = chain_matrix(a, n0)[(size_t)((grp0 + g) * 8 + (c & 7)) * 2 * NP + (row >> 3) * 16 + (row & 7) + (c >= 8 ? 8 : 0)];
                        v = (c >= 8) ? -pv : pv;
                    }
                    else v = a.forcing[(size_t)(a.f_bpr ? n0 + n0 / a.f_bpr : n0) * hstep + (size_t)row * PWc + (grp0 + g) * 16 + c];
                }
                else if (MAT) v = (c < 8 && row == (grp0 + g) * 8 + c) ? 1.0 : 0.0;
                else if (ZERO) v = 0.0;
                else v = a.start[(size_t)b * a.start_stride + (size_t)row * PWc + (grp0 + g) * 16 + c];
            }
            sv[it] = v;
        }
        #pragma unroll
        for (int it = 0; it < NIT; it++) {
            const int e = tid + it * NTH;
            if (e >= NG * NP * 16) continue;
            const int g = e / (NP * 16), el = e % (NP * 16), row = el >> 4, c = el & 15;
            const double v = sv[it];
            part[first][g][el] = v;
            if (MODE == 1 && a.guard_diag && s0 == 0 && !(a.npre > 0 && a.pre_kind[0] == 0)) {   // the window's first point is nobody's product
                const double wv = (row < a.gN) ? a.guard_diag[row + ((c >= 8) ? a.gN : 0)] : 0.0;
                const double trap = (a.n_off == 0) ? 0.5 : 1.0;
                a.guard_forcing[(size_t)row * PWc + (grp0 + g) * 16 + c] = -(2.0 * a.dt / a.tf) * trap * wv * v;
                if (a.count_first) pen += trap * wv * v * v;
            }
        }
    }
    lds_barrier();

#ifdef QGD_CHAIN_PROFILE   // scripts/ubench/chain_bench.hip: clock stamps of block 0 per step
#define CH_STAMP(slot) do { if (blockIdx.x == 0 && rb == 0 && lane == 0 && st < 64) g_chain_prof[st * 8 + (slot)] = clock64(); } while (0)
#else
#define CH_STAMP(slot) do { } while (0)
#endif
    int done = first;                                     // step barriers this wave has passed
    for (int st = st0; st < total; st += NT) {
        while (done < st) { lds_barrier(); done++; }     // steps of the other teams
        CH_STAMP(0);
        const bool mainstep = st >= npfx;
        const int buf = st & 1, n = step_index(mainstep ? st - npfx : 0);
        // C = Are [Bre|Bim] + Aim [-Bim|Bre]: the second right operand is the first with its halves swapped (the LDS address
        // with c16 ^ 8) and a sign on one half.  The sign is the same for every k-step, so the two sums are kept apart and
        // it is applied ONCE, to the finished sum: no vector instruction stands between the MFMAs of a step (a v_xor +
        // v_cndmask in front of every second one cost about a sixth of an MFMA slot each beside the f64 MFMA stream:
        // 77 cycles per MFMA instead of the pipe's 64, scripts/ubench/chain_bench.hip -DQGD_CHAIN_PROFILE).  The B fragments
        // of the whole step are read from LDS up front.
        constexpr int NA = (NG == 1) ? 2 : 1;
        d4 acc[NG][2 * NA];
        #pragma unroll
        for (int g = 0; g < NG; g++)
            #pragma unroll
            for (int q = 0; q < 2 * NA; q++) acc[g][q] = (d4){0, 0, 0, 0};
        double bv1[KST][NG], bv2[KST][NG];
        #pragma unroll
        for (int i = 0; i < KST; i++) {
            const int ko = (i * 4 + kk) * 16;
            #pragma unroll
            for (int g = 0; g < NG; g++) { bv1[i][g] = part[buf][g][ko + c16]; bv2[i][g] = part[buf][g][ko + (c16 ^ 8)]; }
        }
        #pragma unroll
        for (int i = 0; i < KST; i++) {
            #pragma unroll
            for (int g = 0; g < NG; g++) {
                acc[g][2 * (i % NA)] = MFMA(are[i], bv1[i][g], acc[g][2 * (i % NA)]);
                acc[g][2 * (i % NA) + 1] = MFMA(aim[i], bv2[i][g], acc[g][2 * (i % NA) + 1]);
            }
        }
        CH_STAMP(1);
        const int nout = __builtin_amdgcn_readfirstlane(ADJ ? n : n + 1);   // time index of the state this step produces
        double res[NG][4];
        #pragma unroll
        for (int g = 0; g < NG; g++)
            #pragma unroll
            for (int r = 0; r < 4; r++) {
                double vre = acc[g][0][r], vim = acc[g][1][r];
                if (NA == 2) { vre += acc[g][2][r]; vim += acc[g][3][r]; }
                double v = __builtin_fma(sgn2, vim, vre);
                if (FORC) v += fo[g][r];
                res[g][r] = v;
                part[buf ^ 1][g][(rb * 16 + kk + 4 * r) * 16 + c16] = v;
            }
        CH_STAMP(2);
        lds_barrier(); done++;                            // the next team starts; the rest is off the critical path
        CH_STAMP(3);
        if (MODE == 0 && a.mid_every > 0) {              // the running product of the first st+1 matrices, for the sub-block history pass
            const int kdone = st + 1;
            if (kdone >= a.mid_first && kdone < total && (kdone - a.mid_first) % a.mid_every == 0) {
                const size_t pl = (size_t)NP * NP;
                double *pc = a.mid_out + ((size_t)b * a.mid_n + (kdone - a.mid_first) / a.mid_every) * 2 * pl + (c16 >= 8 ? pl : 0);
                #pragma unroll
                for (int g = 0; g < NG; g++)
                    #pragma unroll
                    for (int r = 0; r < 4; r++)
                        pc[(size_t)(rb * 16 + kk + 4 * r) + (size_t)NP * ((grp0 + g) * 8 + (c16 & 7))] = res[g][r];
            }
        }
        if (MODE == 6 && a.mid_out) {                   // (Pi_{e-1} ... Pi_{e-kdone}) = X^H, panel layout, kdone = 2 .. blocks-1
            const int kdone = st + 1;
            if (kdone >= 2 && kdone < total && kdone - 2 < a.mid_n) {
                double *pm = a.mid_out + ((size_t)b * a.mid_n + (kdone - 2)) * 2 * NP * NP;
                #pragma unroll
                for (int g = 0; g < NG; g++)
                    #pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int xrow = rb * 16 + kk + 4 * r, xcol = (grp0 + g) * 8 + (c16 & 7);      // element X[xrow][xcol] (re: c16 < 8, im: else)
                        pm[(size_t)xcol * 2 * NP + (xrow >> 3) * 16 + (xrow & 7) + (c16 >= 8 ? 8 : 0)] = (c16 >= 8) ? -res[g][r] : res[g][r];
                    }
            }
        }
        if (MODE == 2 && a.mid_out) {                   // the affine part of the last kdone blocks of the super-block, kdone = 2 .. blocks-1
            const int kdone = st + 1;
            if (kdone >= 2 && kdone < total && kdone - 2 < a.mid_n) {
                const __amdgpu_buffer_rsrc_t rM = buffer_of(a.mid_out + ((size_t)b * a.mid_n + (kdone - 2)) * hstep + (size_t)grp0 * 16);
                #pragma unroll
                for (int g = 0; g < NG; g++)
                    #pragma unroll
                    for (int r = 0; r < 4; r++)
                        buffer_store_f64(res[g][r], rM, o_lane, (4 * r * PWc + g * 16) * 8);
            }
        }
        if (((MODE == 1 || MODE == 3) && mainstep) || (MODE == 5 && a.out)) {
            const __amdgpu_buffer_rsrc_t rO = buffer_of(a.out + (size_t)nout * hstep + (size_t)grp0 * 16);
            #pragma unroll
            for (int g = 0; g < NG; g++)
                #pragma unroll
                for (int r = 0; r < 4; r++)
                    buffer_store_f64(res[g][r], rO, o_lane, (4 * r * PWc + g * 16) * 8);
        }
        if ((MODE == 1 || MODE == 3) && a.pre_start_out && a.npre > 0 && a.pre_kind[0] == 0 && st == pcnt[0] - 1 &&
            b == (ADJ ? a.nblocks - 1 : 0)) {
            // the state at this window's start (forward) / end (adjoint): nobody's product inside the window
            #pragma unroll
            for (int g = 0; g < NG; g++)
                #pragma unroll
                for (int r = 0; r < 4; r++) {
                    const size_t o = (size_t)(rb * 16 + kk + 4 * r) * PWc + (grp0 + g) * 16 + c16;
                    a.pre_start_out[o] = res[g][r];
                    if (MODE == 1 && a.guard_diag) a.guard_forcing[o] = -(2.0 * a.dt / a.tf) * gw[r] * res[g][r];
                }
        }
        if (MODE == 5 && a.fs_gf) {                      // guard part of dJ/dtheta: -<f_n, s_n>, real stacked form
            #pragma unroll
            for (int g = 0; g < NG; g++) {
                const int cg = (grp0 + g) % a.fs_gpc, PWb = 16 * a.fs_gpc;
                #pragma unroll
                for (int r = 0; r < 4; r++)
                    pen -= a.fs_gf[(size_t)nout * NP * PWb + (size_t)(rb * 16 + kk + 4 * r) * PWb + cg * 16 + c16] * res[g][r];
            }
        }
        if (MODE == 1 && a.guard_diag && mainstep) {
            const double trap = (nout + a.n_off == a.nt_glob - 1) ? 0.5 : 1.0, sc = -(2.0 * a.dt / a.tf) * trap;
            #pragma unroll
            for (int g = 0; g < NG; g++)
                #pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int row = rb * 16 + kk + 4 * r;
                    a.guard_forcing[(size_t)nout * hstep + (size_t)row * PWc + (grp0 + g) * 16 + c16] = sc * gw[r] * res[g][r];
                    pen += trap * gw[r] * res[g][r] * res[g][r];
                }
        }
        CH_STAMP(4);
        if (st + NT < total) issue(st + NT);
        CH_STAMP(5);
    }
    while (done < total) { lds_barrier(); done++; }
    if (MODE == 1 && a.guard_diag) {                     // one atomic per workgroup
        __shared__ double pred[NTH / 16];
        pen = row16_sum(pen);
        if ((lane & 15) == 15) pred[tid >> 4] = pen;
        __syncthreads();
        if (tid == 0) {
            double tot = 0.0;
            for (int q = 0; q < NTH / 16; q++) tot += pred[q];
            if (a.gpart) a.gpart[bid] = tot * a.dt / a.tf; else atomicAdd(&a.scal[2], tot * a.dt / a.tf);
        }
    }
    if (MODE == 5 && a.fs_gf) {                          // NG == 1: one parameter per workgroup
        __shared__ double pred5[NTH / 16];
        pen = row16_sum(pen);
        if ((lane & 15) == 15) pred5[tid >> 4] = pen;
        __syncthreads();
        if (tid == 0) {
            double tot = 0.0;
            for (int q = 0; q < NTH / 16; q++) tot += pred5[q];
            atomicAdd(&a.fs_gacc[grp0 / a.fs_gpc], tot);
        }
    }
    // final state (part[total & 1]) of the block
    if (MODE == 0 || ZERO || (MODE == 5 && a.phi)) {
        const int buf = total & 1;
        for (int e = tid; e < NG * NP * 16; e += NTH) {
            const int g = e / (NP * 16), el = e % (NP * 16), row = el >> 4, c = el & 15;
            const double v = part[buf][g][el];
            if (MODE == 0) {
                const size_t pl = (size_t)NP * NP;
                double *pc = a.PiC + (size_t)b * 2 * pl, *pr = a.PiR + (size_t)b * 2 * pl;
                const int col = (grp0 + g) * 8 + (c & 7);
                pc[(c >= 8 ? pl : 0) + (size_t)row + (size_t)NP * col] = v;
                pr[(size_t)row * 2 * NP + (grp0 + g) * 16 + c] = v;
            } else {
                a.phi[(size_t)b * hstep + (size_t)row * PWc + (grp0 + g) * 16 + c] = v;
            }
        }
    }
}

template <int NP, int MODE, int NG>
__global__ __launch_bounds__(NP * 4 * CHAIN_NT(MODE)) void k_chain_fast(const ChainArgs a)
{
    chain_fast_body<NP, MODE, NG>(a, blockIdx.x, gridDim.x);
}

// Two independent chains in ONE grid (they need the same inputs and nothing from each other): the first nA workgroups
// run MODE_A on `a`, the others MODE_B on `b`.  Used for the level-2 launch of the forward sweep, where the 44 workgroups
// of the super-block products leave most of the chip idle: the suffix products of the adjoint history pass ride along.
template <int NP, int MODE_A, int MODE_B, int NG>
__global__ __launch_bounds__(NP * 4 * CHAIN_NT(MODE_A)) void k_chain_fast2(const ChainArgs a, const ChainArgs b, const int nA)
{
    if ((int)blockIdx.x < nA) chain_fast_body<NP, MODE_A, NG>(a, blockIdx.x, nA);
    else chain_fast_body<NP, MODE_B, NG>(b, (int)blockIdx.x - nA, (int)gridDim.x - nA);
}

// any Np: runtime sizes, each wave loops over its row blocks, no K split, no prefetch
template <int MODE>
__global__ __launch_bounds__(256) void k_chain_generic(const ChainArgs a)
{
    constexpr bool ADJ = (MODE >= 2);
    extern __shared__ double smem[];
    const int Np = a.Np;
    double *cur = smem, *nxt = smem + (size_t)Np * 16;
    int b, grp;
    chain_block_of<MODE>(a, b, grp);
    if (b >= a.nblocks) return;
    const int s0 = b * a.blen, e0 = (s0 + a.blen < a.S) ? s0 + a.blen : a.S;
    const int PWc = (MODE == 0) ? 2 * Np : 2 * a.cp;
    const size_t hstep = (size_t)Np * PWc;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, nw = blockDim.x >> 6;
    const int c16 = lane & 15, kk = lane >> 4;
    for (int e = tid; e < Np * 16; e += blockDim.x) {
        const int row = e >> 4, c = e & 15;
        double v;
        if (MODE == 0) v = (c < 8 && row == grp * 8 + c) ? 1.0 : 0.0;
        else if (MODE == 2) v = 0.0;
        else v = a.start[(size_t)b * a.start_stride + (size_t)row * PWc + grp * 16 + c];
        cur[e] = v;
    }
    __syncthreads();
    for (int st = 0; st < e0 - s0; st++) {
        const int n = ADJ ? e0 - 1 - st : s0 + st;
        const int nout = ADJ ? n : n + 1;
        const double *Pn = chain_matrix(a, n);
        for (int rb = wave; rb * 16 < Np; rb += nw) {
            d4 acc = (d4){0, 0, 0, 0};
            const int arow = rb * 16 + c16;
            for (int k0 = 0; k0 < Np; k0 += 4) {
                double are, aim, b1, b2;
                chain_a<ADJ>(Pn, Np, arow, k0 + kk, are, aim);
                panel_b(cur + (size_t)(k0 + kk) * 16, c16, b1, b2);
                acc = MFMA(are, b1, acc);
                acc = MFMA(aim, b2, acc);
            }
            #pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = rb * 16 + kk + 4 * r;
                double v = acc[r];
                const size_t ho = (size_t)nout * hstep + (size_t)row * PWc + grp * 16 + c16;
                if (ADJ) v += a.forcing[a.f_bpr ? ho + (size_t)(nout / a.f_bpr) * hstep : ho];
                nxt[(size_t)row * 16 + c16] = v;
                if (MODE == 1 || MODE == 3) a.out[ho] = v;
            }
        }
        __syncthreads();
        double *tmp = cur; cur = nxt; nxt = tmp;
    }
    if (MODE == 0) {
        const size_t pl = (size_t)Np * Np;
        double *pc = a.PiC + (size_t)b * 2 * pl, *pr = a.PiR + (size_t)b * 2 * pl;
        for (int e = tid; e < Np * 16; e += blockDim.x) {
            const int row = e >> 4, c = e & 15;
            const int col = grp * 8 + (c & 7);
            pc[(c >= 8 ? pl : 0) + (size_t)row + (size_t)Np * col] = cur[e];
            pr[(size_t)row * 2 * Np + grp * 16 + c] = cur[e];
        }
    }
    if (MODE == 2) {
        for (int e = tid; e < Np * 16; e += blockDim.x)
            a.phi[(size_t)b * hstep + (size_t)(e >> 4) * PWc + grp * 16 + (e & 15)] = cur[e];
    }
}

// MODE 4/5 for any Np (forward sensitivities of eval_grad_forced and eval_forward with a forcing, N > 64 -- the
// reference's cross-check of the adjoint, src/eval_grad_forced.jl:17-194, src/forward_evolution.jl:118-129,167-206, has
// no size limit): the same affine forward chain as k_chain_fast MODE 4/5, one workgroup = (block, column group), the
// state in two LDS panels, no prefetch.  ZERO (MODE 4): start from zero, the block's affine part goes to phi.  MODE 5:
// start from a.start, optional history out[n+1], optional final state to phi, optional guard sums -<f_n, s_n>.
template <int MODE>
__global__ __launch_bounds__(256) void k_chain_forced_generic(const ChainArgs a)
{
    constexpr bool ZERO = (MODE == 4);
    extern __shared__ double smem[];
    __shared__ double pred[16];
    const int Np = a.Np;
    double *cur = smem, *nxt = smem + (size_t)Np * 16;
    const int b = blockIdx.x / a.ngroups, grp = blockIdx.x % a.ngroups;
    if (b >= a.nblocks) return;
    const int s0 = b * a.blen, e0 = (s0 + a.blen < a.S) ? s0 + a.blen : a.S;
    const int PWc = 2 * a.cp;
    const size_t hstep = (size_t)Np * PWc;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, nw = blockDim.x >> 6;
    const int c16 = lane & 15, kk = lane >> 4;
    for (int e = tid; e < Np * 16; e += blockDim.x)
        cur[e] = ZERO ? 0.0 : a.start[(size_t)b * a.start_stride + (size_t)(e >> 4) * PWc + grp * 16 + (e & 15)];
    __syncthreads();
    // the parameter this column group belongs to (fs_mode 1: its forcing is assembled from the basis responses)
    int ctl = 0, lco = 0, ncf = 0;
    const int par = a.fs_gpc ? grp / a.fs_gpc : 0, cg = a.fs_gpc ? grp % a.fs_gpc : 0;
    if (a.fs_mode == 1) {
        while (ctl + 1 < a.fs_nops && par >= a.fs_poff[ctl + 1]) ctl++;
        lco = par - a.fs_poff[ctl]; ncf = a.fs_ncoef[ctl];
    }
    double pen = 0.0;
    for (int st = 0; st < e0 - s0; st++) {
        const int n = s0 + st, nout = n + 1;
        const double *Pn = chain_matrix(a, n);
        for (int rb = wave; rb * 16 < Np; rb += nw) {
            d4 acc = (d4){0, 0, 0, 0};
            const int arow = rb * 16 + c16;
            for (int k0 = 0; k0 < Np; k0 += 4) {
                double are, aim, b1, b2;
                chain_a<false>(Pn, Np, arow, k0 + kk, are, aim);
                panel_b(cur + (size_t)(k0 + kk) * 16, c16, b1, b2);
                acc = MFMA(are, b1, acc);
                acc = MFMA(aim, b2, acc);
            }
            double fo[4] = {0.0, 0.0, 0.0, 0.0};
            if (a.fs_mode == 1) {       // q_n = sum_{tau,d} G^tau_l(n,d) BR[n][b] - G^tau_l(n+1,d) BL[n+1][b]
                const int NB = a.fs_nops * 2 * a.fs_m, PWb = 16 * a.fs_gpc;
                const size_t pstep = (size_t)Np * PWb;
                const double *gk = a.fs_G + a.fs_goff[ctl];
                for (int tau = 0; tau < 2; tau++)
                    for (int d = 0; d < a.fs_m; d++) {
                        const double g0 = gk[(((size_t)tau * a.fs_nt + n + a.fs_n0) * (a.fs_m + 1) + d) * ncf + lco];
                        const double g1 = gk[(((size_t)tau * a.fs_nt + n + a.fs_n0 + 1) * (a.fs_m + 1) + d) * ncf + lco];
                        const int bidx = (ctl * 2 + tau) * a.fs_m + d;
                        const double *br = a.fs_BR + ((size_t)n * NB + bidx) * pstep + cg * 16 + c16;
                        const double *bl = a.fs_BL + ((size_t)(n + 1) * NB + bidx) * pstep + cg * 16 + c16;
                        #pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const size_t ro = (size_t)(rb * 16 + kk + 4 * r) * PWb;
                            fo[r] += g0 * br[ro] - g1 * bl[ro];
                        }
                    }
            } else if (a.forcing) {
                #pragma unroll
                for (int r = 0; r < 4; r++)
                    fo[r] = a.forcing[(size_t)(a.f_bpr ? n + n / a.f_bpr : n) * hstep + (size_t)(rb * 16 + kk + 4 * r) * PWc + grp * 16 + c16];
            }
            #pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = rb * 16 + kk + 4 * r;
                const double v = acc[r] + fo[r];
                nxt[(size_t)row * 16 + c16] = v;
                if (MODE == 5 && a.out) a.out[(size_t)nout * hstep + (size_t)row * PWc + grp * 16 + c16] = v;
                if (MODE == 5 && a.fs_gf) {
                    const int PWb = 16 * a.fs_gpc;
                    pen -= a.fs_gf[(size_t)nout * Np * PWb + (size_t)row * PWb + cg * 16 + c16] * v;
                }
            }
        }
        __syncthreads();
        double *tmp = cur; cur = nxt; nxt = tmp;
    }
    if (MODE == 5 && a.fs_gf) {
        pen = row16_sum(pen);
        if ((lane & 15) == 15) pred[tid >> 4] = pen;
        __syncthreads();
        if (tid == 0) {
            double tot = 0.0;
            for (int q = 0; q < (int)(blockDim.x >> 4); q++) tot += pred[q];
            atomicAdd(&a.fs_gacc[par], tot);
        }
    }
    if (ZERO || (MODE == 5 && a.phi))
        for (int e = tid; e < Np * 16; e += blockDim.x)
            a.phi[(size_t)b * hstep + (size_t)(e >> 4) * PWc + grp * 16 + (e & 15)] = cur[e];
}

// Large N: one workgroup = one (block, tile of 4 column groups = 32 complex columns), all Np rows.
// The step matrix is streamed once per 32 columns instead of once per 8 (the generic kernel is bound by
// exactly that L2 traffic), the state tile lives in ONE LDS buffer [Np][64] (128 KB at Np = 256) and the
// new state in the accumulators until every wave has finished reading the old one.  The 16-wide group
// slot of an LDS row is XOR-ed with row%4 so that the four k-rows of a B fragment fall in different banks.
// 8 waves, wave w owns row blocks w*RBW .. w*RBW+RBW-1; A fragments of step k4+1 are in flight while
// the MFMAs of step k4 issue.  Two barriers per step (a step is >= 25 us of MFMA time at these sizes).
template <int MODE, int RBW, int NGT>
__global__ __launch_bounds__(512) void k_chain_dense(const ChainArgs a)
{
    constexpr bool ADJ = (MODE >= 2);
    extern __shared__ double cur[];
    constexpr int TW = 16 * NGT;                         // doubles per state row of the tile
    const int Np = a.Np, nrb = Np >> 4, ngt = (a.ngroups + NGT - 1) / NGT;
    int b, ct;
    if (MODE == 0) { const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3; b = xcd + 8 * (slot / ngt); ct = slot % ngt; }
    else { b = blockIdx.x / ngt; ct = blockIdx.x % ngt; }
    if (b >= a.nblocks) return;
    const int s0 = b * a.blen, e0 = (s0 + a.blen < a.S) ? s0 + a.blen : a.S;
    const int PWc = (MODE == 0) ? 2 * Np : 2 * a.cp;
    const size_t hstep = (size_t)Np * PWc;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int c16 = lane & 15, kk = lane >> 4;
    // B operand [-Bim | Bre]; the adjoint's A fragment is loaded un-conjugated, the sign goes here
    const int sign_hi = ((c16 < 8) != ADJ) ? (int)0x80000000 : 0;
    for (int e = tid; e < Np * TW; e += blockDim.x) {
        const int row = e / TW, g = (e >> 4) % NGT, c = e & 15, grp = ct * NGT + g;
        double v = 0.0;
        if (grp < a.ngroups) {
            if (MODE == 0) v = (c < 8 && row == grp * 8 + c) ? 1.0 : 0.0;
            else if (MODE != 2) v = a.start[(size_t)b * a.start_stride + (size_t)row * PWc + grp * 16 + c];
        }
        cur[(size_t)row * TW + ((g ^ (row & (NGT - 1))) << 4) + c] = v;
    }
    __syncthreads();
    int rb[RBW];
    #pragma unroll
    for (int r = 0; r < RBW; r++) rb[r] = wave * RBW + r;
    int avr[RBW], avi[RBW];                               // byte offsets of this lane's (row, k = kk) element: chain_a_raw's layouts
    #pragma unroll
    for (int r = 0; r < RBW; r++) {
        const int arow = (rb[r] < nrb ? rb[r] : 0) * 16 + c16;
        if (!ADJ) { avr[r] = (arow + Np * kk) * 8; avi[r] = avr[r] + Np * Np * 8; }
        else { avr[r] = (kk * 2 * Np + (arow >> 3) * 16 + (arow & 7)) * 8; avi[r] = avr[r] + 64; }
    }
    for (int st = 0; st < e0 - s0; st++) {
        const int n = ADJ ? e0 - 1 - st : s0 + st;
        const int nout = ADJ ? n : n + 1;
        const double *Pn = chain_matrix(a, n);
        d4 acc[RBW][NGT];
        #pragma unroll
        for (int r = 0; r < RBW; r++)
            #pragma unroll
            for (int g = 0; g < NGT; g++) acc[r][g] = (d4){0, 0, 0, 0};
        // left operand buffer-addressed (descriptor on the step matrix, per-lane offsets fixed, the k-step in an SGPR); four
        // register sets, three k-steps of prefetch, the loop unrolled by four: no address arithmetic or register copies
        // beside the MFMAs (as k_chain_dense3 below; this kernel serves the 8-column tiles and the sizes 3M would spill at)
        const __amdgpu_buffer_rsrc_t rP = buffer_of(Pn);
        const int kstep = ADJ ? 2 * Np * 8 : Np * 8;           // bytes per unit of k
        double xr[4][RBW], xi[4][RBW];
#define CD4_LOAD(S, kq) do { _Pragma("unroll") for (int r = 0; r < RBW; r++) {                                         \
            xr[S][r] = buffer_load_f64(rP, avr[r], (kq) * kstep); xi[S][r] = buffer_load_f64(rP, avi[r], (kq) * kstep); } } while (0)
#define CD4_STEP(S, kq) do {                                                                                          \
            const double *brow = cur + (size_t)((kq) + kk) * TW + c16;                                                 \
            _Pragma("unroll") for (int g = 0; g < NGT; g++) {                                                          \
                const double b1 = brow[(g ^ (kk & (NGT - 1))) << 4];                                                   \
                const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(b1), 0x128, 0xF, 0xF, false);             \
                const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(b1), 0x128, 0xF, 0xF, false);             \
                const double b2 = __hiloint2double(hi ^ sign_hi, lo);                                                  \
                _Pragma("unroll") for (int r = 0; r < RBW; r++) {                                                      \
                    acc[r][g] = MFMA(xr[S][r], b1, acc[r][g]);                                                         \
                    acc[r][g] = MFMA(xi[S][r], b2, acc[r][g]);                                                         \
                } } } while (0)
        CD4_LOAD(0, 0); CD4_LOAD(1, 4); CD4_LOAD(2, 8);
        for (int k0 = 0; k0 < Np; k0 += 16) {                  // (Np is a multiple of 16)
            const int last = Np - 4;
            CD4_LOAD(3, k0 + 12);                                CD4_STEP(0, k0);
            CD4_LOAD(0, k0 + 16 <= last ? k0 + 16 : last);       CD4_STEP(1, k0 + 4);
            CD4_LOAD(1, k0 + 20 <= last ? k0 + 20 : last);       CD4_STEP(2, k0 + 8);
            CD4_LOAD(2, k0 + 24 <= last ? k0 + 24 : last);       CD4_STEP(3, k0 + 12);
        }
#undef CD4_LOAD
#undef CD4_STEP
        __syncthreads();            // every wave has read the old state
        #pragma unroll
        for (int r = 0; r < RBW; r++) {
            if (rb[r] >= nrb) continue;
            #pragma unroll
            for (int g = 0; g < NGT; g++) {
                const int grp = ct * NGT + g;
                if (grp >= a.ngroups) continue;
                #pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int row = rb[r] * 16 + kk + 4 * e;
                    double v = acc[r][g][e];
                    const size_t ho = (size_t)nout * hstep + (size_t)row * PWc + grp * 16 + c16;
                    if (ADJ) v += a.forcing[a.f_bpr ? ho + (size_t)(nout / a.f_bpr) * hstep : ho];
                    cur[(size_t)row * TW + ((g ^ (kk & (NGT - 1))) << 4) + c16] = v;
                    if (MODE == 1 || MODE == 3) a.out[ho] = v;
                }
            }
        }
        __syncthreads();
    }
    if (MODE == 0 || MODE == 2) {
        const size_t pl = (size_t)Np * Np;
        double *pc = a.PiC + (size_t)b * 2 * pl, *pr = a.PiR + (size_t)b * 2 * pl;
        for (int e = tid; e < Np * TW; e += blockDim.x) {
            const int row = e / TW, g = (e >> 4) % NGT, c = e & 15, grp = ct * NGT + g;
            if (grp >= a.ngroups) continue;
            const double v = cur[(size_t)row * TW + ((g ^ (row & (NGT - 1))) << 4) + c];
            if (MODE == 0) {
                const int col = grp * 8 + (c & 7);
                pc[(c >= 8 ? pl : 0) + (size_t)row + (size_t)Np * col] = v;
                pr[(size_t)row * 2 * Np + grp * 16 + c] = v;
            } else {
                a.phi[(size_t)b * hstep + (size_t)row * PWc + grp * 16 + c] = v;
            }
        }
    }
}

// The same chain with three real products per complex multiply (qgd_k_dense.hip, cgemm3_tile): two column groups form
// one MFMA operand of 16 complex columns; the state tile lies in LDS as [row][pair][16 re | 16 im] (16-double slots,
// XOR-ed with row % NGT as above), so the three right operands are two plain 128-byte rows and their sum.
// Conjugated left operand (ADJ): A^H = ar - i ai with (ar, ai) loaded un-conjugated, hence the sign s below.
//     P1 = ar Br, P2 = ai Bi, P3 = (ar + s ai)(Br + Bi);   Re = P1 - s P2,  Im = P3 - P1 - s P2,   s = +1 / -1 (ADJ).
template <int MODE, int RBW, int NGT>
__global__ __launch_bounds__(512) void k_chain_dense3(const ChainArgs a)
{
    static_assert(NGT == 2 || NGT == 4, "pairs of column groups");
    constexpr bool ADJ = (MODE >= 2);
    constexpr int NPR = NGT / 2;
    extern __shared__ double cur[];
    constexpr int TW = 16 * NGT;                         // doubles per state row of the tile
    const int Np = a.Np, nrb = Np >> 4, ngt = (a.ngroups + NGT - 1) / NGT;
    int b, ct;
    if (MODE == 0) { const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3; b = xcd + 8 * (slot / ngt); ct = slot % ngt; }
    else { b = blockIdx.x / ngt; ct = blockIdx.x % ngt; }
    if (b >= a.nblocks) return;
    const int s0 = b * a.blen, e0 = (s0 + a.blen < a.S) ? s0 + a.blen : a.S;
    const int PWc = (MODE == 0) ? 2 * Np : 2 * a.cp;
    const size_t hstep = (size_t)Np * PWc;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int c16 = lane & 15, kk = lane >> 4;
    // LDS position of element (row, group g of the tile, c in [8 re | 8 im])
    auto lds_at = [&](int row, int g, int c) -> int {
        const int slot = (g >> 1) * 2 + (c >> 3);
        return row * TW + ((slot ^ (row & (NGT - 1))) << 4) + (g & 1) * 8 + (c & 7);
    };
    for (int e = tid; e < Np * TW; e += blockDim.x) {
        const int row = e / TW, g = (e >> 4) % NGT, c = e & 15, grp = ct * NGT + g;
        double v = 0.0;
        if (grp < a.ngroups) {
            if (MODE == 0) v = (c < 8 && row == grp * 8 + c) ? 1.0 : 0.0;
            else if (MODE != 2) v = a.start[(size_t)b * a.start_stride + (size_t)row * PWc + grp * 16 + c];
        }
        cur[lds_at(row, g, c)] = v;
    }
    __syncthreads();
    int rb[RBW];
    #pragma unroll
    for (int r = 0; r < RBW; r++) rb[r] = wave * RBW + r;
    const double sgn = ADJ ? -1.0 : 1.0;
    int avr[RBW], avi[RBW];                               // byte offsets of this lane's (row, k = kk) element: chain_a_raw's layouts
    #pragma unroll
    for (int r = 0; r < RBW; r++) {
        const int arow = (rb[r] < nrb ? rb[r] : 0) * 16 + c16;
        if (!ADJ) { avr[r] = (arow + Np * kk) * 8; avi[r] = avr[r] + Np * Np * 8; }
        else { avr[r] = (kk * 2 * Np + (arow >> 3) * 16 + (arow & 7)) * 8; avi[r] = avr[r] + 64; }
    }
    for (int st = 0; st < e0 - s0; st++) {
        const int n = ADJ ? e0 - 1 - st : s0 + st;
        const int nout = ADJ ? n : n + 1;
        const double *Pn = chain_matrix(a, n);
        d4 p1[RBW][NPR], p2[RBW][NPR], p3[RBW][NPR];
        #pragma unroll
        for (int r = 0; r < RBW; r++)
            #pragma unroll
            for (int p = 0; p < NPR; p++) { p1[r][p] = (d4){0, 0, 0, 0}; p2[r][p] = (d4){0, 0, 0, 0}; p3[r][p] = (d4){0, 0, 0, 0}; }
        // left operand: buffer-addressed (descriptor on the step matrix, per-lane offsets fixed, the k-step in an SGPR)
        const __amdgpu_buffer_rsrc_t rP = buffer_of(Pn);
        const int kstep = ADJ ? 2 * Np * 8 : Np * 8;           // bytes per unit of k
        // four register sets, three k-steps of prefetch, the loop unrolled by four: no register copies, constant LDS offsets
        double xr[4][RBW], xi[4][RBW];
#define CD3_LOAD(S, kq) do { _Pragma("unroll") for (int r = 0; r < RBW; r++) {                                         \
            xr[S][r] = buffer_load_f64(rP, avr[r], (kq) * kstep); xi[S][r] = buffer_load_f64(rP, avi[r], (kq) * kstep); } } while (0)
#define CD3_STEP(S, kq) do {                                                                                          \
            const double *brow = cur + (size_t)((kq) + kk) * TW + c16;                                                 \
            double as[RBW];                                                                                           \
            _Pragma("unroll") for (int r = 0; r < RBW; r++) as[r] = ADJ ? xr[S][r] - xi[S][r] : xr[S][r] + xi[S][r];     \
            _Pragma("unroll") for (int p = 0; p < NPR; p++) {                                                          \
                const double bre = brow[((2 * p) ^ (kk & (NGT - 1))) << 4], bim = brow[((2 * p + 1) ^ (kk & (NGT - 1))) << 4]; \
                const double bs = bre + bim;                                                                          \
                _Pragma("unroll") for (int r = 0; r < RBW; r++) {                                                      \
                    p1[r][p] = MFMA(xr[S][r], bre, p1[r][p]);                                                          \
                    p2[r][p] = MFMA(xi[S][r], bim, p2[r][p]);                                                          \
                    p3[r][p] = MFMA(as[r], bs, p3[r][p]);                                                              \
                } } } while (0)
        CD3_LOAD(0, 0); CD3_LOAD(1, 4); CD3_LOAD(2, 8);
        for (int k0 = 0; k0 < Np; k0 += 16) {                  // (Np is a multiple of 16)
            const int last = Np - 4;
            CD3_LOAD(3, k0 + 12);                                CD3_STEP(0, k0);
            CD3_LOAD(0, k0 + 16 <= last ? k0 + 16 : last);       CD3_STEP(1, k0 + 4);
            CD3_LOAD(1, k0 + 20 <= last ? k0 + 20 : last);       CD3_STEP(2, k0 + 8);
            CD3_LOAD(2, k0 + 24 <= last ? k0 + 24 : last);       CD3_STEP(3, k0 + 12);
        }
#undef CD3_LOAD
#undef CD3_STEP
        __syncthreads();            // every wave has read the old state
        #pragma unroll
        for (int r = 0; r < RBW; r++) {
            if (rb[r] >= nrb) continue;
            #pragma unroll
            for (int p = 0; p < NPR; p++) {
                const int g = 2 * p + (c16 >> 3), grp = ct * NGT + g;
                if (grp >= a.ngroups) continue;
                #pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int row = rb[r] * 16 + kk + 4 * e;
                    const double sp2 = sgn * p2[r][p][e];
                    double vre = p1[r][p][e] - sp2, vim = p3[r][p][e] - p1[r][p][e] - sp2;
                    const size_t ho = (size_t)nout * hstep + (size_t)row * PWc + grp * 16 + (c16 & 7);
                    if (ADJ) {
                        const size_t fo = a.f_bpr ? ho + (size_t)(nout / a.f_bpr) * hstep : ho;
                        vre += a.forcing[fo]; vim += a.forcing[fo + 8];
                    }
                    cur[(size_t)row * TW + (((2 * p) ^ (kk & (NGT - 1))) << 4) + c16] = vre;
                    cur[(size_t)row * TW + (((2 * p + 1) ^ (kk & (NGT - 1))) << 4) + c16] = vim;
                    if (MODE == 1 || MODE == 3) { a.out[ho] = vre; a.out[ho + 8] = vim; }
                }
            }
        }
        __syncthreads();
    }
    if (MODE == 0 || MODE == 2) {
        const size_t pl = (size_t)Np * Np;
        double *pc = a.PiC + (size_t)b * 2 * pl, *pr = a.PiR + (size_t)b * 2 * pl;
        for (int e = tid; e < Np * TW; e += blockDim.x) {
            const int row = e / TW, g = (e >> 4) % NGT, c = e & 15, grp = ct * NGT + g;
            if (grp >= a.ngroups) continue;
            const double v = cur[lds_at(row, g, c)];
            if (MODE == 0) {
                const int col = grp * 8 + (c & 7);
                pc[(c >= 8 ? pl : 0) + (size_t)row + (size_t)Np * col] = v;
                pr[(size_t)row * 2 * Np + grp * 16 + c] = v;
            } else {
                a.phi[(size_t)b * hstep + (size_t)row * PWc + grp * 16 + c] = v;
            }
        }
    }
}

// Tile width: 4 column groups per workgroup stream the step matrix once per 32 columns -- the right choice when
// there are more tiles than CUs (config 5: 32 blocks x 8 tiles).  When there are not, the step time of ONE workgroup
// is what counts and narrower tiles put more CUs on the same chain (N = 100, 32 columns, 54 blocks: one 4-group tile
// per block is 54 workgroups at 10.5 us per step, four 1-group tiles are 216 at 1.5 us).
template <int MODE, int NGT>
static int launch_chain_dense_w(const ChainArgs &a, hipStream_t stream)
{
    const int ngt = (a.ngroups + NGT - 1) / NGT, nrb = a.Np / 16;
    const int nwg = (MODE == 0) ? 8 * ngt * ((a.nblocks + 7) / 8) : a.nblocks * ngt;
    const size_t shm = (size_t)a.Np * 16 * NGT * sizeof(double);
    if (nwg <= 0) return 0;
    const bool m3 = qgd_path("dense_4m") == nullptr;          // three-product tiles (k_chain_dense3) unless switched off (tests)
#define CALL_CD(R) do { bool done3_ = false;                                                                                \
        if constexpr (NGT > 1 && !(NGT == 4 && R == 3) && R < 5) {    /* (<.,3,4>, <.,5,2>: spills -- the four-product kernel there) */ \
            if (m3) { SET_LDS_ONCE((k_chain_dense3<MODE, R, NGT>), shm);                                 \
            hipLaunchKernelGGL((k_chain_dense3<MODE, R, NGT>), dim3(nwg), dim3(512), shm, stream, a); done3_ = true; } }      \
        if (!done3_) { SET_LDS_ONCE((k_chain_dense<MODE, R, NGT>), shm);                                                     \
            hipLaunchKernelGGL((k_chain_dense<MODE, R, NGT>), dim3(nwg), dim3(512), shm, stream, a); } } while (0)
    if (nrb <= 8) CALL_CD(1); else if (nrb <= 16) CALL_CD(2); else if (nrb <= 24) CALL_CD(3);
    else if constexpr (NGT <= 2) { if (nrb <= 32) CALL_CD(4); else CALL_CD(5); }      // (288 < Np <= 640: tiles of at most 16 columns fit the LDS)
    else return -1;
#undef CALL_CD
    return (int)hipGetLastError();
}

// A single sequential chain of states (nblocks = 1: the states at the super-block starts, the prefix over the lower ranks'
// windows) on a big problem: one full-chip GEMM launch per step instead of one chain kernel on cp/8 workgroups
// (k_chain_step in qgd_k_dense.hip; config 5: 28 -> 10 us per dependent step).  MODE 1: cur = P_n cur, out[n+1] = cur;
// MODE 3: cur = P_n^H cur + forcing[n], out[n] = cur.
template <int MODE>
static bool chain_as_steps(const ChainArgs &a)
{
    if (MODE != 1 && MODE != 3) return false;
    if (a.nblocks != 1 || a.S < 1 || qgd_path("dense_4m")) return false;
    if (a.fs_mode || a.guard_diag || a.sub_T || a.npre) return false;                  // plain chains only
    const char *e = qgd_path("chain_steps_min");                                      // (tests: the step path on small shapes)
    return (long long)a.Np * a.Np * a.cp >= (e ? atoll(e) : 256LL * 256 * 64);
}

template <int MODE>
static int launch_chain_steps(const ChainArgs &a, hipStream_t stream)
{
    constexpr bool ADJ = (MODE == 3);
    const size_t hstep = (size_t)a.Np * 2 * a.cp;
    const int e0 = (a.blen < a.S) ? a.blen : a.S;
    const double *cur = a.start;
    for (int st = 0; st < e0; st++) {
        const int n = ADJ ? e0 - 1 - st : st, nout = ADJ ? n : n + 1;
        const double *f = nullptr;
        if (ADJ) f = a.forcing + (size_t)(a.f_bpr ? nout + nout / a.f_bpr : nout) * hstep;
        double *o = a.out + (size_t)nout * hstep;
        const int rc = qgdk_dense_chain_step(stream, ADJ ? 1 : 0, chain_matrix(a, n), cur, o, f, a.Np, a.cp);
        if (rc) return rc;
        cur = o;
    }
    return 0;
}

template <int MODE>
static int launch_chain_dense(const ChainArgs &a, hipStream_t stream)
{
    if (chain_as_steps<MODE>(a)) return launch_chain_steps<MODE>(a, stream);
    const long long cus = 256;      // (thresholds of 128 .. 1024 tiles give the same times within 2 %)
    const long long t1 = cus - 1, t2 = 2 * cus;     // (8-column tiles only while they are fewer than the CUs)
    if ((long long)a.nblocks * a.ngroups <= t1) return launch_chain_dense_w<MODE, 1>(a, stream);
    if ((long long)a.nblocks * ((a.ngroups + 1) / 2) <= t2 || a.Np > 288) return launch_chain_dense_w<MODE, 2>(a, stream);     // (32-column tiles: Np <= 288, LDS)
    return launch_chain_dense_w<MODE, 4>(a, stream);
}

// the dense chain pays when a tile of 4 groups is (nearly) full and the state tile fits in LDS
static bool chain_is_dense(const ChainArgs &a)
{
    const bool off = qgd_path("chain_generic") != nullptr;       // (tests: the generic chain, which Np > 640 takes, on a small shape)
    return !off && a.Np > 64 && a.Np <= 640;
}

template <int MODE, int NG>
static int launch_chain_ng(const ChainArgs &a, hipStream_t stream)
{
    const bool fast = (a.Np == 16 || a.Np == 32 || a.Np == 48 || a.Np == 64) && a.ngroups % NG == 0;
    const int ng = fast ? a.ngroups / NG : a.ngroups;
    const int nwg = ((MODE == 0) ? 8 * ng * ((a.nblocks + 7) / 8) : a.nblocks * ng) + ((MODE == 2 && fast && a.t_on) ? 1 : 0) +
                    ((MODE == 0 && fast && a.p0_on) ? 1 : 0);
    if (nwg <= 0) return 0;
    switch (fast ? a.Np : 0) {
    case 16: hipLaunchKernelGGL((k_chain_fast<16, MODE, NG>), dim3(nwg), dim3(16 * 4 * CHAIN_NT(MODE)), 0, stream, a); break;
    case 32: hipLaunchKernelGGL((k_chain_fast<32, MODE, NG>), dim3(nwg), dim3(32 * 4 * CHAIN_NT(MODE)), 0, stream, a); break;
    case 48: hipLaunchKernelGGL((k_chain_fast<48, MODE, NG>), dim3(nwg), dim3(48 * 4 * CHAIN_NT(MODE)), 0, stream, a); break;
    case 64:         hipLaunchKernelGGL((k_chain_fast<64, MODE, NG>), dim3(nwg), dim3(64 * 4 * CHAIN_NT(MODE)), 0, stream, a); break;
    default: {
        if (MODE >= 4) {
            const size_t shm4 = (size_t)2 * a.Np * 16 * sizeof(double);
            if (shm4 > 64 * 1024) HIPCHK(hipFuncSetAttribute((const void *)k_chain_forced_generic<(MODE >= 4 ? MODE : 4)>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm4));
            hipLaunchKernelGGL((k_chain_forced_generic<(MODE >= 4 ? MODE : 4)>), dim3(a.nblocks * a.ngroups), dim3(256), shm4, stream, a);
            return (int)hipGetLastError();
        }
        if (chain_is_dense(a)) return launch_chain_dense<(MODE >= 4 ? 1 : MODE)>(a, stream);
        size_t shm = (size_t)2 * a.Np * 16 * sizeof(double);
        hipLaunchKernelGGL((k_chain_generic<(MODE >= 4 ? 1 : MODE)>), dim3(nwg), dim3(256), shm, stream, a);
    }
    }
    return (int)hipGetLastError();
}

template <int MODE>
static int launch_chain(const ChainArgs &a, hipStream_t stream)
{
    // the matrix-matrix chains (block propagators) pair up column groups when one group per
    // workgroup would need more than one workgroup per CU: half the L2 traffic for the step
    // matrices at the same MFMA time; small grids keep one group per workgroup (shorter steps)
    if (MODE == 0 && a.ngroups % 2 == 0 && (long long)a.nblocks * a.ngroups > 256) return launch_chain_ng<MODE, 2>(a, stream);   // (one group per workgroup, two workgroups per CU: the same within 2 %)
    return launch_chain_ng<MODE, 1>(a, stream);
}

// level-2 products of the forward sweep (MODE 0 over the block propagators) with the suffix products of the adjoint
// history pass (MODE 6) in the same grid; falls back to the plain MODE 0 launch where MODE 6 does not apply
static int launch_chain_level2(const ChainArgs &a0, const ChainArgs &a6, bool with_suffix, hipStream_t stream)
{
    const bool fast = (a0.Np == 16 || a0.Np == 32 || a0.Np == 48 || a0.Np == 64);
    if (!with_suffix || !fast || (long long)a0.nblocks * a0.ngroups > 256) return launch_chain<0>(a0, stream);
    const int nA = 8 * a0.ngroups * ((a0.nblocks + 7) / 8) + (a0.p0_on ? 1 : 0), nB = 8 * a6.ngroups * ((a6.nblocks + 7) / 8);
    switch (a0.Np) {
    case 16: hipLaunchKernelGGL((k_chain_fast2<16, 0, 6, 1>), dim3(nA + nB), dim3(16 * 4 * CHAIN_NT(0)), 0, stream, a0, a6, nA); break;
    case 32: hipLaunchKernelGGL((k_chain_fast2<32, 0, 6, 1>), dim3(nA + nB), dim3(32 * 4 * CHAIN_NT(0)), 0, stream, a0, a6, nA); break;
    case 48: hipLaunchKernelGGL((k_chain_fast2<48, 0, 6, 1>), dim3(nA + nB), dim3(48 * 4 * CHAIN_NT(0)), 0, stream, a0, a6, nA); break;
    default: hipLaunchKernelGGL((k_chain_fast2<64, 0, 6, 1>), dim3(nA + nB), dim3(64 * 4 * CHAIN_NT(0)), 0, stream, a0, a6, nA); break;
    }
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------
// K5: guard penalty and adjoint forcing (infidelity.jl:56-96,
// eval_grad_discrete_adjoint.jl:732-752).  One workgroup per time point.
//   f_n = -(2 dt/tf) * trap_n * W w_n ;  penalty += (dt/tf) trap_n w_n^T W w_n
// W: dense real 2N x 2N, column-major (unpadded).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_guard(const double *__restrict__ W,
                                               const double *__restrict__ hist,
                                               double *__restrict__ forcing,
                                               double *__restrict__ scal, int N, int Np, int c,
                                               int cp, int n_off, int nt_glob, int count_first, double dt, double tf,
                                               int have_guard, double *__restrict__ gpart)
{
    __shared__ double red[4];
    const int n = blockIdx.x, ng = n + n_off;
    const int PWc = 2 * cp;
    const size_t hstep = (size_t)Np * PWc;
    const double *h = hist + (size_t)n * hstep;
    double *f = forcing + (size_t)n * hstep;
    const double trap = (ng == 0 || ng == nt_glob - 1) ? 0.5 : 1.0;
    double pen = 0.0;
    if (!have_guard) {
        for (int e = threadIdx.x; e < (int)hstep; e += blockDim.x) f[e] = 0.0;
        return;
    }
    // zero the padding
    for (int e = threadIdx.x; e < (int)hstep; e += blockDim.x) f[e] = 0.0;
    __syncthreads();
    const int n2 = 2 * N;
    for (int e = threadIdx.x; e < n2 * c; e += blockDim.x) {
        const int i = e % n2, col = e / n2;
        const int cbase = (col >> 3) * 16 + (col & 7);
        double s = 0.0;
        for (int jx = 0; jx < n2; jx++) {
            const double wj = (jx < N) ? h[(size_t)jx * PWc + cbase] : h[(size_t)(jx - N) * PWc + cbase + 8];
            s += W[(size_t)i + (size_t)n2 * jx] * wj;
        }
        const size_t o = (i < N) ? (size_t)i * PWc + cbase : (size_t)(i - N) * PWc + cbase + 8;
        pen += h[o] * s;
        f[o] = -(2.0 * dt / tf) * trap * s;
    }
    for (int off = 32; off > 0; off >>= 1) pen += __shfl_down(pen, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = pen;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double tot = (n > 0 || count_first) ? (red[0] + red[1] + red[2] + red[3]) * trap * dt / tf : 0.0;
        if (gpart) gpart[n] = tot;                       // added in index order by the terminal stage
        else if (n > 0 || count_first) atomicAdd(&scal[2], tot);
    }
}

// K5 (fast path): W diagonal (the guard_projector of multi_qudit_systems.jl:316-349 is):
// elementwise forcing and penalty.  wd: the 2N diagonal entries.
__global__ __launch_bounds__(256) void k_guard_diag(const double *__restrict__ wd,
                                                    const double *__restrict__ hist,
                                                    double *__restrict__ forcing,
                                                    double *__restrict__ scal, int N, int Np, int cp,
                                                    int n_off, int nt_glob, int count_first, double dt, double tf, double *__restrict__ gpart)
{
    __shared__ double red[4];
    const int n = blockIdx.x, ng = n + n_off;     // local / global time index
    const int PWc = 2 * cp;
    const size_t hstep = (size_t)Np * PWc;
    const double *h = hist + (size_t)n * hstep;
    double *f = forcing + (size_t)n * hstep;
    const double trap = (ng == 0 || ng == nt_glob - 1) ? 0.5 : 1.0;
    const double sc = -(2.0 * dt / tf) * trap;
    double pen = 0.0;
    for (int e = threadIdx.x; e < (int)hstep; e += blockDim.x) {
        const int row = e / PWc, c16 = (e % PWc) & 15;
        const double w = (row < N) ? wd[row + ((c16 >= 8) ? N : 0)] : 0.0;
        const double v = h[e];
        f[e] = sc * w * v;
        pen += w * v * v;
    }
    for (int off = 32; off > 0; off >>= 1) pen += __shfl_down(pen, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = pen;
    __syncthreads();
    // the first point of a time window is the last point of the previous rank's window
    if (threadIdx.x == 0) {
        const double tot = (n > 0 || count_first) ? (red[0] + red[1] + red[2] + red[3]) * trap * dt / tf : 0.0;
        if (gpart) gpart[n] = tot;                       // added in index order by the terminal stage
        else if (n > 0 || count_first) atomicAdd(&scal[2], tot);
    }
}

// ---------------------------------------------------------------------------
// K6: overlaps and terminal right-hand side (infidelity.jl:13-17,
// eval_grad_discrete_adjoint.jl:22-40).  Single workgroup.
//   scal[0] = <w_N,R>, scal[1] = <w_N,T>;  y_N = (2/Ness^2)(a R + b T) + f_N
// ---------------------------------------------------------------------------
// the overlaps and the terminal condition of the adjoint (infidelity.jl:7-18; eval_grad_discrete_adjoint.jl:1-67 in the
// variable y = L^T lambda): one workgroup, any size up to 1024 threads
__device__ __forceinline__ void terminal_block(const double *__restrict__ hist, const double *__restrict__ target,
                                               const double *__restrict__ forcing, double *__restrict__ yhist,
                                               double *__restrict__ scal, int Np, int cp, int nt, int n_ess, int have_target,
                                               int write_y, double *__restrict__ y2, double *__restrict__ y3,
                                               double *__restrict__ y4, int given_ab, const double *gpart, int gpart_n,
                                               const double *ytarget)
{
    // ytarget (fused front, qgd_front.h): the terminal value is lambda_N = L_N^-H rhs_N itself -- the overlaps are taken with
    // `target`, the state is formed from ytarget = L_N^-H target (:Tracking / :Norm: L_N^-H rhs) and `forcing` = h = L^-H f
    __shared__ double red[32];
    __shared__ double gred[16];
    const int PWc = 2 * cp, nw = blockDim.x >> 6;          // 4 waves, or 16 for large panels
    const size_t hstep = (size_t)Np * PWc;
    const double *w = hist + (size_t)(nt - 1) * hstep;
    // have_target: 0 none, 1 :Infidelity, 2 :Tracking, 3 :Norm (eval_grad_discrete_adjoint.jl:26-35).  The last two are
    // local in the columns: scal[0] = 0.5 |w_N - R|^2 (0.5 |w_N|^2), scal[1] = 0, y_N = -(w_N - R) + f_N (-w_N + f_N).
    const int cost = have_target > 1 ? have_target - 1 : 0;
    double a = 0.0, b = 0.0;
    if (cost) {
        for (int e = threadIdx.x; e < (int)hstep; e += blockDim.x) {
            const double d = (cost == 1) ? w[e] - target[e] : w[e];
            a += 0.5 * d * d;
        }
    } else if (have_target && !given_ab) {
        for (int e = threadIdx.x; e < (int)hstep; e += blockDim.x) {
            const int col = e % PWc;
            const int c16 = col & 15;
            const double tv = target[e];
            a += w[e] * tv;
            // T = [R_im; -R_re]: pairs u with R_im and v with -R_re
            const double tp = target[e ^ 8];           // partner (re<->im) of the same column
            b += (c16 < 8) ? w[e] * tp : -w[e] * tp;
        }
    }
    // guard penalty: the partial sums of the guard stage's workgroups, added in a FIXED order (thread t takes the entries
    // t, t + blockDim, ... in ascending order; then the same shuffle tree and wave order as the overlaps) -- the same bits
    // on every run: the objective an optimizer compares from step to step is infidelity + this number
    double gs = 0.0;
    if (gpart) for (int w = threadIdx.x; w < gpart_n; w += blockDim.x) gs += gpart[w];
    for (int off = 32; off > 0; off >>= 1) { a += __shfl_down(a, off); b += __shfl_down(b, off); gs += __shfl_down(gs, off); }
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = a; red[16 + (threadIdx.x >> 6)] = b; gred[threadIdx.x >> 6] = gs; }
    __syncthreads();
    a = 0.0; b = 0.0;
    for (int q = 0; q < nw; q++) { a += red[q]; b += red[16 + q]; }
    if (gpart && threadIdx.x == 0) { gs = 0.0; for (int q = 0; q < nw; q++) gs += gred[q]; scal[2] = gs; }
    // column shards: the overlaps are GLOBAL sums over all columns (infidelity.jl:13-17); after the ranks' all-reduce
    // they are in scal and the terminal condition is formed from them, not from this rank's columns
    if (given_ab) { a = scal[0]; b = scal[1]; }       // (a column shard's Tracking / Norm cost was reduced like the overlaps)
    else if (threadIdx.x == 0) { scal[0] = a; scal[1] = b; }
    if (!write_y) return;
    if (cost) {
        double *yc = yhist + (size_t)(nt - 1) * hstep;
        const double *fc = forcing + (size_t)(nt - 1) * hstep;
        for (int e = threadIdx.x; e < (int)hstep; e += blockDim.x) {
            const double v = (ytarget ? ytarget[e] : ((cost == 1) ? target[e] - w[e] : -w[e])) + fc[e];      // (fused front: ytarget = L_N^-H rhs, k_psi)
            yc[e] = v; y2[e] = v; y3[e] = v; y4[e] = v;
        }
        return;
    }
    const double sc = 2.0 / ((double)n_ess * (double)n_ess);
    double *y = yhist + (size_t)(nt - 1) * hstep;
    const double *f = forcing + (size_t)(nt - 1) * hstep;
    if (ytarget) target = ytarget;
    for (int e = threadIdx.x; e < (int)hstep; e += blockDim.x) {
        const int c16 = (e % PWc) & 15;
        const double tv = target[e], tp = target[e ^ 8];
        const double Tv = (c16 < 8) ? tp : -tp;        // T component at this slot
        const double v = sc * (a * tv + b * Tv) + f[e];
        y[e] = v; y2[e] = v; y3[e] = v; y4[e] = v;    // history, exchange slot, boundary arrays
    }
}

// ---------------------------------------------------------------------------
// K8: lambda_n = Linv_n^H y_n for n = 1..nt-1 (parallel over n and column groups)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_lambda(const double *__restrict__ LinvT,
                                                const double *__restrict__ yhist,
                                                double *__restrict__ lam, int Np, int cp,
                                                double *__restrict__ zero_a, int n_a,
                                                double *__restrict__ zero_b, int n_b)
{
    {   // the gradient kernels accumulate into sigma and grad: clear them here (saves two fills)
        const int gid = (blockIdx.y * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x;
        const int gsz = gridDim.x * gridDim.y * blockDim.x;
        for (int e = gid; e < n_a; e += gsz) zero_a[e] = 0.0;
        for (int e = gid; e < n_b; e += gsz) zero_b[e] = 0.0;
    }
    extern __shared__ double smem[];
    double *ys = smem;                                   // [Np][16]
    const int n = blockIdx.y + 1, grp = blockIdx.x;
    const int PWc = 2 * cp;
    const size_t hstep = (size_t)Np * PWc, pl = (size_t)Np * Np;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    const int c16 = lane & 15, kk = lane >> 4;
    for (int e = threadIdx.x; e < Np * 16; e += blockDim.x)
        ys[e] = yhist[(size_t)n * hstep + (size_t)(e >> 4) * PWc + grp * 16 + (e & 15)];
    __syncthreads();
    const double *Tre = LinvT + (size_t)n * 2 * pl, *Tim = Tre + pl;   // (r,c) at c + Np*r
    for (int rb = wave; rb * 16 < Np; rb += nw) {
        d4 acc = (d4){0, 0, 0, 0};
        const int arow = rb * 16 + c16;
        for (int k0 = 0; k0 < Np; k0 += 4) {
            const int k = k0 + kk;
            // (Linv^H)(row,k) = conj(Linv(k,row)); Linv(k,row) sits at row + Np*k
            const double are = Tre[(size_t)arow + (size_t)Np * k];
            const double aim = -Tim[(size_t)arow + (size_t)Np * k];
            double b1, b2;
            panel_b(ys + (size_t)k * 16, c16, b1, b2);
            acc = MFMA(are, b1, acc);
            acc = MFMA(aim, b2, acc);
        }
        #pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = rb * 16 + kk + 4 * r;
            lam[(size_t)n * hstep + (size_t)row * PWc + grp * 16 + c16] = acc[r];
        }
    }
}

// the same with the size at compile time (Np <= 64): the 2 * Np/4 fragment loads of a wave are all issued before the
// first MFMA needs one (the run-time loop above waits for a global round trip per k-step)
template <int NPC>
__global__ __launch_bounds__(256) void k_lambda_c(const double *__restrict__ LinvT, const double *__restrict__ yhist,
                                                  double *__restrict__ lam, int cp, double *__restrict__ zero_a, int n_a,
                                                  double *__restrict__ zero_b, int n_b)
{
    {
        const int gid = (blockIdx.y * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x;
        const int gsz = gridDim.x * gridDim.y * blockDim.x;
        for (int e = gid; e < n_a; e += gsz) zero_a[e] = 0.0;
        for (int e = gid; e < n_b; e += gsz) zero_b[e] = 0.0;
    }
    __shared__ double ys[NPC * 16];
    const int n = blockIdx.y + 1, grp = blockIdx.x;
    const int PWc = 2 * cp;
    const size_t hstep = (size_t)NPC * PWc, pl = (size_t)NPC * NPC;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c16 = lane & 15, kk = lane >> 4;
    const double *Tre = LinvT + (size_t)n * 2 * pl, *Tim = Tre + pl;
    double are[NPC / 4], aim[NPC / 4];
    const bool live = wave * 16 < NPC;
    if (live) {
        const int arow = wave * 16 + c16;
        #pragma unroll
        for (int i = 0; i < NPC / 4; i++) { are[i] = Tre[arow + NPC * (4 * i + kk)]; aim[i] = Tim[arow + NPC * (4 * i + kk)]; }
    }
    for (int e = threadIdx.x; e < NPC * 16; e += blockDim.x)
        ys[e] = yhist[(size_t)n * hstep + (size_t)(e >> 4) * PWc + grp * 16 + (e & 15)];
    __syncthreads();
    if (!live) return;
    d4 acc0 = (d4){0, 0, 0, 0}, acc1 = (d4){0, 0, 0, 0};
    #pragma unroll
    for (int i = 0; i < NPC / 4; i++) {
        double b1, b2;
        panel_b(ys + (size_t)(4 * i + kk) * 16, c16, b1, b2);
        acc0 = MFMA(are[i], b1, acc0);
        acc1 = MFMA(-aim[i], b2, acc1);          // (Linv^H)(row,k) = conj(Linv(k,row))
    }
    #pragma unroll
    for (int r = 0; r < 4; r++)
        lam[(size_t)n * hstep + (size_t)(wave * 16 + kk + 4 * r) * PWc + grp * 16 + c16] = acc0[r] + acc1[r];
}


// ---------------------------------------------------------------------------
// Fused front (qgd_front.h), between the two sweeps: per (column group, time point)
//   psi_n = L_n^-1 phi_n = X_n^H phi_n                      (the state history; psi_0 is the given initial state)
//   f_n = -(2 dt/tf) trap_n W psi_n, the guard penalty        (eval_grad_discrete_adjoint.jl:732-752, infidelity.jl:56-96; W diagonal)
//   h_n = L_n^-H f_n = X_n f_n                               (the forcing of the adjoint sweep in lambda)
//   n = nt-1: termU = X_N target                            (terminal_block forms lambda_N from it)
// X_n = L_n^-H: row-major planes in LinvT, read once and staged through LDS plane by plane (below).
// ---------------------------------------------------------------------------
typedef double psi_d2 __attribute__((ext_vector_type(2)));
#ifdef QGD_STAMPS
__device__ unsigned long long g_stamps_chain[1024][8];
extern "C" int qgdk_stamps_chain(unsigned long long *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps_chain), sizeof(g_stamps_chain)); }
#define PSI_STAMP(i) do { if (blockIdx.x < 1024 && threadIdx.x == 0) g_stamps_chain[blockIdx.x][i] = wall_clock64(); } while (0)
#else
#define PSI_STAMP(i) do { } while (0)
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_psi(const double *__restrict__ LinvT, const double *__restrict__ phist, const double *__restrict__ psi0,
                                             double *__restrict__ hist, double *__restrict__ forcing, double *__restrict__ hforc,
                                             const double *__restrict__ guard_diag, double *__restrict__ gpart,
                                             const double *__restrict__ target, double *__restrict__ termU,
                                             const int cp, const int nt, const int gN, const double dt, const double tf, const int cost)
{
    constexpr int NPC = 64, XS = 68;
    __shared__ __attribute__((aligned(16))) double xs[NPC * XS];      // one plane of X_n at a time, then the right operands
    double *bs = xs;
    __shared__ double red[16];
    const int n = blockIdx.x, grp = blockIdx.y;      // (time point fastest: the workgroups of a CU are then three different time points)
    const int PWc = 2 * cp;
    const size_t hstep = (size_t)NPC * PWc, pl = (size_t)NPC * NPC;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c16 = lane & 15, kk = lane >> 4, arow = wave * 16 + c16;
    const double *Tre = LinvT + (size_t)n * 2 * pl, *Tim = Tre + pl;
    const bool guard = guard_diag != nullptr, last = (n == nt - 1) && target != nullptr;
    // X_n is read from memory ONCE (both left operands read straight from global memory --
    // the second one with the lanes of a row block 512 bytes apart -- took 26 us for the 551 time points of the headline
    // against 11.6 for k_lambda_c's single pass: 4.4 MB of L^-H per XCD do not stay in its 4 MB of L2 between the two).
    // Each plane goes through LDS, where X^H (16 consecutive rows of one k) and X (16 consecutive k of one row) are both
    // fragment reads.
    // (every load instruction of a wave reads 1 KB contiguous: thread t takes the 16-byte pieces t, t + 256, ... of a plane)
    PSI_STAMP(0);
    double xr[16], xi[16];
    #pragma unroll
    for (int q = 0; q < 8; q++) {
        const int e2 = 2 * (q * 256 + (int)threadIdx.x);
        const psi_d2 vr = *reinterpret_cast<const psi_d2 *>(Tre + e2), vi = *reinterpret_cast<const psi_d2 *>(Tim + e2);
        xr[2 * q] = vr[0]; xr[2 * q + 1] = vr[1]; xi[2 * q] = vi[0]; xi[2 * q + 1] = vi[1];
    }
    const double *src = (n > 0) ? phist + (size_t)n * hstep : psi0;
    double bv[4];
    #pragma unroll
    for (int q = 0; q < 4; q++) { const int e = threadIdx.x + 256 * q; bv[q] = src[(size_t)(e >> 4) * PWc + grp * 16 + (e & 15)]; }
    double are[NPC / 4], aim[NPC / 4], are2[NPC / 4], aim2[NPC / 4];
    #pragma unroll
    for (int q = 0; q < 8; q++) {
        const int e2 = 2 * (q * 256 + (int)threadIdx.x);
        *reinterpret_cast<psi_d2 *>(xs + (e2 >> 6) * XS + (e2 & 63)) = (psi_d2){xr[2 * q], xr[2 * q + 1]};
    }
    __syncthreads();
    #pragma unroll
    for (int i = 0; i < NPC / 4; i++) { are[i] = xs[(4 * i + kk) * XS + arow]; are2[i] = xs[arow * XS + 4 * i + kk]; }
    __syncthreads();
    #pragma unroll
    for (int q = 0; q < 8; q++) {
        const int e2 = 2 * (q * 256 + (int)threadIdx.x);
        *reinterpret_cast<psi_d2 *>(xs + (e2 >> 6) * XS + (e2 & 63)) = (psi_d2){xi[2 * q], xi[2 * q + 1]};
    }
    __syncthreads();
    #pragma unroll
    for (int i = 0; i < NPC / 4; i++) { aim[i] = xs[(4 * i + kk) * XS + arow]; aim2[i] = xs[arow * XS + 4 * i + kk]; }
    __syncthreads();                                     // the planes are in registers: their space takes the right operand
    #pragma unroll
    for (int q = 0; q < 4; q++) bs[threadIdx.x + 256 * q] = bv[q];
    __syncthreads();
    PSI_STAMP(1);
    double res[4];
    if (n > 0) {
        d4 acc0 = (d4){0, 0, 0, 0}, acc1 = (d4){0, 0, 0, 0};
        #pragma unroll
        for (int i = 0; i < NPC / 4; i++) {
            double b1, b2;
            panel_b(bs + (size_t)(4 * i + kk) * 16, c16, b1, b2);
            acc0 = MFMA(are[i], b1, acc0);
            acc1 = MFMA(-aim[i], b2, acc1);          // (X^H)(row,k) = conj(X(k,row))
        }
        #pragma unroll
        for (int r = 0; r < 4; r++) {
            res[r] = acc0[r] + acc1[r];
            hist[(size_t)n * hstep + (size_t)(wave * 16 + kk + 4 * r) * PWc + grp * 16 + c16] = res[r];
        }
    } else {
        #pragma unroll
        for (int r = 0; r < 4; r++) res[r] = bs[(wave * 16 + kk + 4 * r) * 16 + c16];
    }
    PSI_STAMP(2);
    if (!guard && !last) return;
    if (guard) {
        const double trap = (n == 0 || n == nt - 1) ? 0.5 : 1.0, sc = -(2.0 * dt / tf) * trap;
        double pen = 0.0, fv[4];
        #pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = wave * 16 + kk + 4 * r;
            const double gw = (row < gN) ? guard_diag[row + ((c16 >= 8) ? gN : 0)] : 0.0;
            fv[r] = sc * gw * res[r];
            pen += gw * res[r] * res[r];
            forcing[(size_t)n * hstep + (size_t)row * PWc + grp * 16 + c16] = fv[r];
        }
        __syncthreads();                                 // every wave has read the first right operand
        #pragma unroll
        for (int r = 0; r < 4; r++) bs[(wave * 16 + kk + 4 * r) * 16 + c16] = fv[r];
        pen = row16_sum(pen);
        if (c16 == 15) red[threadIdx.x >> 4] = pen;
        __syncthreads();
        if (threadIdx.x == 0) {
            double tot = 0.0;
            #pragma unroll
            for (int q = 0; q < 16; q++) tot += red[q];
            gpart[(size_t)n * gridDim.y + grp] = tot * trap * dt / tf;      // added in index order by the terminal stage
        }
        d4 acc0 = (d4){0, 0, 0, 0}, acc1 = (d4){0, 0, 0, 0};
        #pragma unroll
        for (int i = 0; i < NPC / 4; i++) {
            double b1, b2;
            panel_b(bs + (size_t)(4 * i + kk) * 16, c16, b1, b2);
            acc0 = MFMA(are2[i], b1, acc0);
            acc1 = MFMA(aim2[i], b2, acc1);
        }
        #pragma unroll
        for (int r = 0; r < 4; r++)
            hforc[(size_t)n * hstep + (size_t)(wave * 16 + kk + 4 * r) * PWc + grp * 16 + c16] = acc0[r] + acc1[r];
        PSI_STAMP(3);
    }
    if (last) {
        // cost 0 (:Infidelity): U = X_N target, lambda_N = (2/N_ess^2)(a + ib) U + h_N; :Tracking / :Norm: the terminal right-hand
        // side is local, -(psi_N - target) / -psi_N (eval_grad_discrete_adjoint.jl:26-35): U = X_N rhs, lambda_N = U + h_N
        __syncthreads();
        if (cost == 0) {
            for (int e = threadIdx.x; e < NPC * 16; e += 256) bs[e] = target[(size_t)(e >> 4) * PWc + grp * 16 + (e & 15)];
        } else {
            #pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = wave * 16 + kk + 4 * r;
                const double tv = (cost == 1) ? target[(size_t)row * PWc + grp * 16 + c16] : 0.0;
                bs[row * 16 + c16] = tv - res[r];
            }
        }
        __syncthreads();
        d4 acc0 = (d4){0, 0, 0, 0}, acc1 = (d4){0, 0, 0, 0};
        #pragma unroll
        for (int i = 0; i < NPC / 4; i++) {
            double b1, b2;
            panel_b(bs + (size_t)(4 * i + kk) * 16, c16, b1, b2);
            acc0 = MFMA(are2[i], b1, acc0);
            acc1 = MFMA(aim2[i], b2, acc1);
        }
        #pragma unroll
        for (int r = 0; r < 4; r++) termU[(size_t)(wave * 16 + kk + 4 * r) * PWc + grp * 16 + c16] = acc0[r] + acc1[r];
    }
}

// the same in two launches of many workgroups, for panels of >= 32768 elements: partial overlaps per workgroup, then y_N.
// The partial sums are STORED (part[2 b], part[2 b + 1]); the workgroup that draws the last ticket adds them in index order
// and writes scal[0..1] -- the same bits whichever workgroup that is (the gradient depends on these two numbers through
// the terminal condition: with atomicAdd here the N > 64 gradient differed in its last bits from run to run).
__global__ __launch_bounds__(256) void k_terminal_sum(const double *__restrict__ w, const double *__restrict__ target,
                                                      double *__restrict__ scal, int hstep, int PWc, int cost, double *__restrict__ part, int chunk,
                                                      const double *__restrict__ gpart, int gpart_n)
{
    __shared__ double red[8];
    __shared__ int last;
    double a = 0.0, b = 0.0;
    for (int e = blockIdx.x * chunk + threadIdx.x; e < min(hstep, (int)(blockIdx.x + 1) * chunk); e += 256) {
        if (cost) { const double d = (cost == 1) ? w[e] - target[e] : w[e]; a += 0.5 * d * d; continue; }   // :Tracking / :Norm
        const int c16 = (e % PWc) & 15;
        const double tp = target[e ^ 8];
        a += w[e] * target[e];
        b += (c16 < 8) ? w[e] * tp : -w[e] * tp;
    }
    for (int off = 32; off > 0; off >>= 1) { a += __shfl_down(a, off); b += __shfl_down(b, off); }
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = a; red[4 + (threadIdx.x >> 6)] = b; }
    __syncthreads();
    int *ticket = reinterpret_cast<int *>(part + 2 * 1024);
    if (threadIdx.x == 0) {
        part[2 * blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
        part[2 * blockIdx.x + 1] = ((red[4] + red[5]) + red[6]) + red[7];
        __threadfence();                                          // release: the partial sums before the ticket
        last = (atomicAdd(ticket, 1) == (int)gridDim.x - 1) ? 1 : 0;
    }
    __syncthreads();
    if (!last || threadIdx.x != 0) return;
    __threadfence();                                              // acquire: every other workgroup's partial sums
    double A = 0.0, B = 0.0;
    for (int g = 0; g < (int)gridDim.x; g++) { A += __builtin_nontemporal_load(part + 2 * g); B += __builtin_nontemporal_load(part + 2 * g + 1); }
    scal[0] = A; scal[1] = B;
    if (gpart) { double s = 0.0; for (int w = 0; w < gpart_n; w++) s += gpart[w]; scal[2] = s; }      // guard penalty, in index order
    *ticket = 0;                                                  // for the next evaluation (same stream: ordered)
}

__global__ __launch_bounds__(256) void k_terminal_y(const double *__restrict__ target, const double *__restrict__ f,
                                                    const double *__restrict__ scal, double *__restrict__ y,
                                                    double *__restrict__ y2, double *__restrict__ y3, double *__restrict__ y4,
                                                    int hstep, int PWc, int n_ess, int cost, const double *__restrict__ w)
{
    const double a = scal[0], b = scal[1], sc = 2.0 / ((double)n_ess * (double)n_ess);
    for (int e = blockIdx.x * 2048 + threadIdx.x; e < min(hstep, (int)(blockIdx.x + 1) * 2048); e += 256) {
        if (cost) { const double v = ((cost == 1) ? target[e] - w[e] : -w[e]) + f[e]; y[e] = v; y2[e] = v; y3[e] = v; y4[e] = v; continue; }
        const int c16 = (e % PWc) & 15;
        const double tv = target[e], tp = target[e ^ 8];
        const double Tv = (c16 < 8) ? tp : -tp;
        const double v = sc * (a * tv + b * Tv) + f[e];
        y[e] = v; y2[e] = v; y3[e] = v; y4[e] = v;
    }
}

__global__ __launch_bounds__(1024) void k_terminal(const double *__restrict__ hist,
                                                  const double *__restrict__ target,
                                                  const double *__restrict__ forcing,
                                                  double *__restrict__ yhist,
                                                  double *__restrict__ scal, int Np, int cp, int nt,
                                                  int n_ess, int have_target, int write_y,
                                                  double *__restrict__ y2, double *__restrict__ y3,
                                                  double *__restrict__ y4, int given_ab, const double *__restrict__ gpart, int gpart_n,
                                                  const double *__restrict__ ytarget)
{
    terminal_block(hist, target, forcing, yhist, scal, Np, cp, nt, n_ess, have_target, write_y, y2, y3, y4, given_ab, gpart, gpart_n, ytarget);
}

// scal[2] += the guard partials in a fixed order (thread-strided ascending sums, shuffle tree, waves in order): where no
// terminal stage follows the guard stage -- a window of a long grid, a rank that does not own the final time
__global__ __launch_bounds__(256) void k_guard_fold(const double *__restrict__ gpart, int n, double *__restrict__ scal)
{
    __shared__ double red[4];
    double gs = 0.0;
    for (int w = threadIdx.x; w < n; w += 256) gs += gpart[w];
    for (int off = 32; off > 0; off >>= 1) gs += __shfl_down(gs, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = gs;
    __syncthreads();
    if (threadIdx.x == 0) scal[2] += ((red[0] + red[1]) + red[2]) + red[3];
}

extern "C" {

#define QGD_SUB_LEN 3      /* steps per sub-block of the forward history pass */
static inline bool chain_is_fast(const qgdk_ctx *c) { return c->Np == 16 || c->Np == 32 || c->Np == 48 || c->Np == 64; }

// diagonal guard projector + compiled-size sweeps: k_chain_fast<.,1,.> does the guard work
static inline bool guard_is_fused(const qgdk_ctx *c)
{
    return !c->front && c->have_guard == 2 && (c->Np == 16 || c->Np == 32 || c->Np == 48 || c->Np == 64);
}

// ---------------------------------------------------------------------------
// The scan over time has three levels.  A rank owns a window of B = bpr blocks (level 1: blen steps
// each); inside the window the block boundaries are reached by a second level (B2 super-blocks of g
// blocks) when B > 8; across ranks only the product of a whole window, R_r = Pi_{B-1} ... Pi_0
// (forward) and its affine part phi^rank_r (adjoint), are exchanged -- 128 KB + 16 KB per rank --
// and every rank runs the short chain over the windows before / after its own.
//   PiX  : local  [B x Pi planes | B x Pi panel]          phiX : local [B x phi]
//   RX   : exchange buffer 0, per rank [R planes | R panel]
//   phiRX: exchange buffer 1, per rank [phi^rank | y_N (last rank only)]
// ---------------------------------------------------------------------------
// the adjoint history pass takes the blocks after its own inside the super-block in ONE step (stored suffix products and
// affine parts, ChainArgs::suf_P): compiled-size chains with a second scan level of more than two blocks per super-block
static inline bool suffix_on(const qgdk_ctx *c)
{
    return c->SufP && c->SufPhi && chain_is_fast(c) && c->scan_blocks2 > 1 && c->scan_g > 2 && !qgd_path("no_suffix");
}

static inline size_t rx_chunk(const qgdk_ctx *c) { return (size_t)4 * c->Np * c->Np; }
static inline size_t phirx_chunk(const qgdk_ctx *c) { return (size_t)2 * c->Np * 2 * c->cp; }

// forward, part 1 (no other rank needed): block propagators, super-block propagators, window product
int qgdk_forward_blocks(const qgdk_ctx *c)
{
    const size_t pl2 = (size_t)2 * c->Np * c->Np;
    const int B = c->scan_blocks, B2 = c->scan_blocks2, g = c->scan_g;
    int rc;
    ChainArgs a{};
    a.Np = c->Np; a.cp = c->cp; a.S = c->nt - 1; a.Pmat = c->Pc;
    a.PiC = c->PiX; a.PiR = c->PiX + (size_t)B * pl2;
    a.nblocks = B; a.blen = c->scan_blen; a.ngroups = c->Np / 8;
    if (c->sub_hist) { a.mid_out = c->Hmid; a.mid_every = QGD_SUB_LEN; a.mid_first = QGD_SUB_LEN; a.mid_n = c->sub_n; }
    // fused front: phi_0 = L_0 psi_0 rides along -- in the level-2 launch where there is one (most of the chip is idle there;
    // beside the block products, one workgroup per CU, the extra workgroup cost the launch 2.4 us), else here
    if (c->front && B2 <= 1) { a.p0_on = 1; a.p0_E = c->L; a.p0_psi0 = c->psi0; a.p0_out = c->phi0; }
    if ((rc = launch_chain<0>(a, c->stream))) return rc;
    if (B2 > 1) {      // super-block propagators from the block propagators
        ChainArgs a2{};
        a2.Np = c->Np; a2.cp = c->cp; a2.S = B; a2.Pmat = c->PiX;
        a2.PiC = c->PiC2; a2.PiR = c->PiR2; a2.nblocks = B2; a2.blen = g; a2.ngroups = c->Np / 8;
        if (c->front) { a2.p0_on = 1; a2.p0_E = c->L; a2.p0_psi0 = c->psi0; a2.p0_out = c->phi0; }
        if (c->sub_hist && g > 2) { a2.mid_out = c->Qmid; a2.mid_every = 1; a2.mid_first = 2; a2.mid_n = g - 2; }
        // beside them (same grid, idle CUs): the suffix products of the blocks of every super-block for the adjoint history pass
        ChainArgs a6{};
        const bool suf = suffix_on(c);
        if (suf) {
            a6.Np = c->Np; a6.cp = c->cp; a6.S = B; a6.Pmat = c->PiX + (size_t)B * pl2; a6.nblocks = B2; a6.blen = g; a6.ngroups = c->Np / 8;
            a6.mid_out = c->SufP; a6.mid_n = g - 2;
        }
        if ((rc = launch_chain_level2(a2, a6, suf, c->stream))) return rc;
    }
    if (c->part_world > 1) {   // the product of the whole window, into this rank's chunk of RX
        ChainArgs r{};
        r.Np = c->Np; r.cp = c->cp; r.ngroups = c->Np / 8; r.nblocks = 1;
        if (B2 > 1) { r.S = B2; r.blen = B2; r.Pmat = c->PiC2; } else { r.S = B; r.blen = B; r.Pmat = c->PiX; }
        r.PiC = c->RX + (size_t)c->part_rank * rx_chunk(c); r.PiR = r.PiC + pl2;
        if ((rc = launch_chain<0>(r, c->stream))) return rc;
    }
    return 0;
}

// forward, part 1 in pieces (ranges of blocks): the block propagators of blocks [b0, b1) on `stream`;
// then the levels above them (super-blocks, window product) on the context's stream
int qgdk_forward_blocks_range(const qgdk_ctx *c, int b0, int b1, hipStream_t stream)
{
    const size_t pl2 = (size_t)2 * c->Np * c->Np;
    const int B = c->scan_blocks, S = c->nt - 1;
    const int s_lo = b0 * c->scan_blen, s_hi = (b1 * c->scan_blen < S) ? b1 * c->scan_blen : S;
    if (b1 <= b0 || s_hi <= s_lo) return 0;
    ChainArgs a{};
    a.Np = c->Np; a.cp = c->cp; a.S = s_hi - s_lo; a.Pmat = c->Pc + (size_t)s_lo * pl2;
    a.PiC = c->PiX + (size_t)b0 * pl2; a.PiR = c->PiX + ((size_t)B + b0) * pl2;
    a.nblocks = b1 - b0; a.blen = c->scan_blen; a.ngroups = c->Np / 8;
    if (c->sub_hist) { a.mid_out = c->Hmid + (size_t)b0 * c->sub_n * pl2; a.mid_every = QGD_SUB_LEN; a.mid_first = QGD_SUB_LEN; a.mid_n = c->sub_n; }
    return launch_chain<0>(a, stream);
}

int qgdk_forward_blocks_upper(const qgdk_ctx *c)
{
    const size_t pl2 = (size_t)2 * c->Np * c->Np;
    const int B = c->scan_blocks, B2 = c->scan_blocks2, g = c->scan_g;
    int rc;
    if (B2 > 1) {      // super-block propagators from the block propagators
        ChainArgs a2{};
        a2.Np = c->Np; a2.cp = c->cp; a2.S = B; a2.Pmat = c->PiX;
        a2.PiC = c->PiC2; a2.PiR = c->PiR2; a2.nblocks = B2; a2.blen = g; a2.ngroups = c->Np / 8;
        if (c->sub_hist && g > 2) { a2.mid_out = c->Qmid; a2.mid_every = 1; a2.mid_first = 2; a2.mid_n = g - 2; }
        // beside them (same grid, idle CUs): the suffix products of the blocks of every super-block for the adjoint history pass
        ChainArgs a6{};
        const bool suf = suffix_on(c);
        if (suf) {
            a6.Np = c->Np; a6.cp = c->cp; a6.S = B; a6.Pmat = c->PiX + (size_t)B * pl2; a6.nblocks = B2; a6.blen = g; a6.ngroups = c->Np / 8;
            a6.mid_out = c->SufP; a6.mid_n = g - 2;
        }
        if ((rc = launch_chain_level2(a2, a6, suf, c->stream))) return rc;
    }
    if (c->part_world > 1) {   // the product of the whole window, into this rank's chunk of RX
        ChainArgs r{};
        r.Np = c->Np; r.cp = c->cp; r.ngroups = c->Np / 8; r.nblocks = 1;
        if (B2 > 1) { r.S = B2; r.blen = B2; r.Pmat = c->PiC2; } else { r.S = B; r.blen = B; r.Pmat = c->PiX; }
        r.PiC = c->RX + (size_t)c->part_rank * rx_chunk(c); r.PiR = r.PiC + pl2;
        if ((rc = launch_chain<0>(r, c->stream))) return rc;
    }
    return 0;
}

// forward, part 2 (after the all-gather of RX): state at the window start, at the super-block and block
// starts, then the history of the own blocks
int qgdk_forward_finish(const qgdk_ctx *c)
{
    const size_t hstep = (size_t)c->Np * 2 * c->cp;
    const int B = c->scan_blocks, B2 = c->scan_blocks2, g = c->scan_g;
    int rc;
    if (chain_is_fast(c)) {
        // one launch: the workgroup of block b first advances psi_0 over the windows of the lower ranks, the
        // super-blocks and the blocks before b (prefix segments of k_chain_fast), then writes its history
        ChainArgs s3{};
        s3.Np = c->Np; s3.cp = c->cp; s3.S = c->nt - 1; s3.Pmat = c->Pc; s3.start = c->psi0; s3.start_stride = 0;
        s3.out = c->hist; s3.nblocks = B; s3.blen = c->scan_blen; s3.ngroups = c->cp / 8;
        if (c->front) { s3.start = c->phi0; s3.out = c->phist; }      // fused front: the sweep runs in phi = L psi (k_psi follows)
        int q = 0;
        if (c->part_rank > 0) {
            s3.pre_kind[q] = 0; s3.pre_P[q] = c->RX; s3.pre_pm_bpr[q] = 1; s3.pre_pm_chunk[q] = (long long)rx_chunk(c);
            s3.pre_rank_count = c->part_rank; s3.pre_start_out = c->hist; q++;
        }
        if (B2 > 1) { s3.pre_kind[q] = 1; s3.pre_P[q] = c->PiC2; q++; }
        if (B > 1) { s3.pre_kind[q] = 2; s3.pre_P[q] = c->PiX; q++; }
        s3.npre = q; s3.pre_g = g; s3.pre_B2 = B2;
        if (c->sub_hist) {   // sub-blocks of QGD_SUB_LEN steps on their own workgroups, prefixes through the stored products
            s3.sub_T = c->sub_n + 1; s3.sub_len = QGD_SUB_LEN; s3.sub_n = c->sub_n; s3.sub_H = c->Hmid;
            s3.nblocks = B * s3.sub_T;
            if (B2 > 1 && g > 2) { s3.pre_Q = c->Qmid; s3.pre_Qn = g - 2; }
        }
        if (guard_is_fused(c)) {
            s3.guard_diag = c->guard_diag; s3.guard_forcing = c->forcing; s3.scal = c->scal; s3.gN = c->N;
            s3.gpart = c->gpart_on ? c->gpart : nullptr;
            s3.n_off = c->n_off; s3.nt_glob = c->nt_glob; s3.count_first = (c->n_off == 0) ? 1 : 0; s3.dt = c->dt; s3.tf = c->tf;
        }
        return launch_chain<1>(s3, c->stream);
    }
    if (c->part_rank > 0) {   // psi at the window start = R_{r-1} ... R_0 psi_0
        ChainArgs w{};
        w.Np = c->Np; w.cp = c->cp; w.S = c->part_rank; w.Pmat = c->RX; w.pm_bpr = 1; w.pm_chunk = (long long)rx_chunk(c);
        w.start = c->psi0; w.start_stride = 0; w.out = c->wbnd; w.nblocks = 1; w.blen = c->part_rank; w.ngroups = c->cp / 8;
        if ((rc = launch_chain<1>(w, c->stream))) return rc;
        const double *ws = c->wbnd + (size_t)c->part_rank * hstep;
        HIPCHK(hipMemcpyAsync(c->bnd, ws, hstep * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->bnd2, ws, hstep * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->hist, ws, hstep * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    }   // rank 0: bnd[0] = bnd2[0] = hist[0] = psi_0 were written when the grid was allocated
    if (B2 <= 1) {
        ChainArgs s2{};
        s2.Np = c->Np; s2.cp = c->cp; s2.S = B; s2.Pmat = c->PiX;
        s2.start = c->bnd; s2.start_stride = 0; s2.out = c->bnd; s2.nblocks = 1; s2.blen = B; s2.ngroups = c->cp / 8;
        if ((rc = launch_chain<1>(s2, c->stream))) return rc;
    } else {
        ChainArgs b2{};   // states at super-block starts
        b2.Np = c->Np; b2.cp = c->cp; b2.S = B2; b2.Pmat = c->PiC2; b2.start = c->bnd2; b2.start_stride = 0; b2.out = c->bnd2;
        b2.nblocks = 1; b2.blen = B2; b2.ngroups = c->cp / 8;
        if ((rc = launch_chain<1>(b2, c->stream))) return rc;
        ChainArgs c2{};   // states at every block start
        c2.Np = c->Np; c2.cp = c->cp; c2.S = B; c2.Pmat = c->PiX;
        c2.start = c->bnd2; c2.start_stride = (long long)hstep; c2.out = c->bnd; c2.nblocks = B2; c2.blen = g; c2.ngroups = c->cp / 8;
        if ((rc = launch_chain<1>(c2, c->stream))) return rc;
    }
    ChainArgs s3{};
    s3.Np = c->Np; s3.cp = c->cp; s3.S = c->nt - 1; s3.Pmat = c->Pc; s3.start = c->bnd;
    s3.start_stride = (long long)hstep; s3.out = c->hist; s3.nblocks = B; s3.blen = c->scan_blen;
    s3.ngroups = c->cp / 8;
    if (guard_is_fused(c)) {
        s3.guard_diag = c->guard_diag; s3.guard_forcing = c->forcing; s3.scal = c->scal; s3.gN = c->N;
        s3.gpart = c->gpart_on ? c->gpart : nullptr;
        s3.n_off = c->n_off; s3.nt_glob = c->nt_glob; s3.count_first = (c->n_off == 0) ? 1 : 0; s3.dt = c->dt; s3.tf = c->tf;
    }
    return launch_chain<1>(s3, c->stream);
}

int qgdk_guard_is_fused(const qgdk_ctx *c) { return guard_is_fused(c) ? 1 : 0; }
int qgdk_guard_fold(const qgdk_ctx *c)
{
    hipLaunchKernelGGL(k_guard_fold, dim3(1), dim3(256), 0, c->stream, c->gpart, c->gpart_n, c->scal);
    return (int)hipGetLastError();
}
// workgroups of the guard stage (= entries of gpart it writes): the history pass when the guard work is fused into it
// (chain_is_fast sizes: one per (block or sub-block, column group)), else one per time point (k_guard_diag / k_guard)
int qgdk_guard_parts(const qgdk_ctx *c)
{
    if (c->front) return c->nt * (c->cp / 8);        // k_psi: one per (time point, column group)
    if (!guard_is_fused(c)) return c->nt;
    const int nb = c->sub_hist ? c->scan_blocks * (c->sub_n + 1) : c->scan_blocks;
    return nb * (c->cp / 8);
}
int qgdk_terminal_can_fuse(const qgdk_ctx *c) { return chain_is_fast(c) && !qgd_path("terminal_kernel"); }

int qgdk_guard(const qgdk_ctx *c)
{
    if (guard_is_fused(c)) return 0;          // the history pass of the forward sweep wrote forcing and penalty
    return qgdk_guard_kernel(c);
}

// the stand-alone guard kernels (also used after a forced forward sweep, whose history pass has no guard part)
int qgdk_guard_kernel(const qgdk_ctx *c)
{
    const int count_first = (c->n_off == 0) ? 1 : 0;
    if (c->have_guard == 2) {   // diagonal projector
        hipLaunchKernelGGL(k_guard_diag, dim3(c->nt), dim3(256), 0, c->stream, c->guard_diag, c->hist, c->forcing,
                           c->scal, c->N, c->Np, c->cp, c->n_off, c->nt_glob, count_first, c->dt, c->tf, c->gpart_on ? c->gpart : nullptr);
        return (int)hipGetLastError();
    }
    hipLaunchKernelGGL(k_guard, dim3(c->nt), dim3(256), 0, c->stream, c->guard, c->hist, c->forcing, c->scal,
                       c->N, c->Np, c->c, c->cp, c->n_off, c->nt_glob, count_first, c->dt, c->tf, c->have_guard,
                       c->gpart_on ? c->gpart : nullptr);
    return (int)hipGetLastError();
}

static int launch_terminal(const qgdk_ctx *c, int write_y, int given_ab);
int qgdk_terminal(const qgdk_ctx *c, int write_y) { return launch_terminal(c, write_y, 0); }
int qgdk_terminal_given(const qgdk_ctx *c) { return launch_terminal(c, 1, 1); }
static int launch_terminal(const qgdk_ctx *c, int write_y, int given_ab)
{
    const size_t hstep = (size_t)c->Np * 2 * c->cp;
    double *slot = c->phiRX + (size_t)c->part_rank * phirx_chunk(c) + hstep;     // y_N for the other ranks
    if (hstep >= 32768) {
        // large panels (config 5: 131072 elements): one workgroup took 177 us; many workgroups, two launches
        const int nwg = (int)((hstep + 2047) / 2048);
        const int tchunk = 2048 * ((nwg + 1023) / 1024), twg = (int)((hstep + tchunk - 1) / tchunk);      // (<= 1024 partial sums)
        const double *w = c->hist + (size_t)(c->nt - 1) * hstep;
        if (!given_ab) {
            HIPCHK(hipMemsetAsync(c->scal, 0, 2 * sizeof(double), c->stream));
            if (c->have_target) hipLaunchKernelGGL(k_terminal_sum, dim3(twg), dim3(256), 0, c->stream, w, c->target, c->scal, (int)hstep, 2 * c->cp, c->cost_type, c->term_part, tchunk,
                                                   (c->gpart_on && c->gpart_terminal && c->have_guard) ? c->gpart : nullptr, c->gpart_n);
            else if (c->gpart_on && c->gpart_terminal && c->have_guard)      // no target: only the guard penalty is to be added up
                hipLaunchKernelGGL(k_terminal, dim3(1), dim3(256), 0, c->stream, c->hist, c->target, c->forcing, c->yhist, c->scal, c->Np, c->cp, c->nt,
                                   c->n_ess, 0, 0, slot, slot, slot, 0, c->gpart, c->gpart_n, (const double *)nullptr);
        }
        if (write_y)
            hipLaunchKernelGGL(k_terminal_y, dim3(nwg), dim3(256), 0, c->stream, c->target, c->forcing + (size_t)(c->nt - 1) * hstep,
                               c->scal, c->yhist + (size_t)(c->nt - 1) * hstep, slot, c->bndY + (size_t)c->scan_blocks * hstep,
                               c->bndY2 + (size_t)c->scan_blocks2 * hstep, (int)hstep, 2 * c->cp, c->n_ess, c->cost_type, w);
        return (int)hipGetLastError();
    }
    // (fused front: the terminal value is lambda_N, formed from termU = L_N^-H target and h_N, into lambda's history)
    hipLaunchKernelGGL(k_terminal, dim3(1), dim3(hstep >= 32768 ? 1024 : 256), 0, c->stream, c->hist, c->target, c->front ? c->hforc : c->forcing,
                       c->front ? c->lam : c->yhist,
                       c->scal, c->Np, c->cp, c->nt, c->n_ess, c->have_target * (1 + c->cost_type), write_y, slot,
                       c->bndY + (size_t)c->scan_blocks * hstep, c->bndY2 + (size_t)c->scan_blocks2 * hstep, given_ab,
                       (c->gpart_on && c->gpart_terminal && c->have_guard && !given_ab) ? c->gpart : nullptr, c->gpart_n,
                       c->front ? c->termU : (const double *)nullptr);
    return (int)hipGetLastError();
}

// adjoint, part 1 (no other rank needed): affine parts of the blocks, of the super-blocks, of the window
int qgdk_adjoint_blocks(const qgdk_ctx *c)
{
    const size_t pl2 = (size_t)2 * c->Np * c->Np;
    const int B = c->scan_blocks, B2 = c->scan_blocks2, g = c->scan_g;
    const double *PiRx = c->PiX + (size_t)B * pl2;       // panel copies of the block propagators
    int rc;
    ChainArgs a{};
    a.Np = c->Np; a.cp = c->cp; a.S = c->nt - 1; a.Pmat = c->Pr; a.forcing = c->front ? c->hforc : c->forcing; a.phi = c->phiX;
    a.nblocks = B; a.blen = c->scan_blen; a.ngroups = c->cp / 8;
    if (c->fuse_terminal && chain_is_fast(c)) {          // y_N and the overlaps by an extra workgroup of this launch
        const size_t hstep = (size_t)c->Np * 2 * c->cp;
        a.t_on = 1; a.t_nt = c->nt; a.t_ness = c->n_ess; a.t_have_target = c->have_target * (1 + c->cost_type);
        a.t_hist = c->hist; a.t_target = c->target; a.t_forcing = c->forcing; a.t_yhist = c->yhist; a.t_scal = c->scal;
        if (c->front) { a.t_forcing = c->hforc; a.t_yhist = c->lam; a.t_ytarget = c->termU; }      // fused front: lambda_N itself, into lambda's history
        a.t_gpart = (c->gpart_on && c->gpart_terminal && c->have_guard) ? c->gpart : nullptr; a.t_gpart_n = c->gpart_n;
        a.t_y2 = c->phiRX + (size_t)c->part_rank * phirx_chunk(c) + hstep;
        a.t_y3 = c->bndY + (size_t)c->scan_blocks * hstep; a.t_y4 = c->bndY2 + (size_t)c->scan_blocks2 * hstep;
    }
    if ((rc = launch_chain<2>(a, c->stream))) return rc;
    if (B2 > 1) {      // affine parts of the super-blocks (their propagators PiR2 come from the forward sweep)
        ChainArgs a2{};
        a2.Np = c->Np; a2.cp = c->cp; a2.S = B; a2.Pmat = PiRx; a2.forcing = c->phiX; a2.phi = c->phi2;
        a2.nblocks = B2; a2.blen = g; a2.ngroups = c->cp / 8;
        if (suffix_on(c)) { a2.mid_out = c->SufPhi; a2.mid_n = g - 2; }      // running affine parts: the suffix sums of the history pass
        if ((rc = launch_chain<2>(a2, c->stream))) return rc;
    }
    if (c->part_world > 1) {   // affine part of the whole window, into this rank's chunk of phiRX
        ChainArgs r{};
        r.Np = c->Np; r.cp = c->cp; r.ngroups = c->cp / 8; r.nblocks = 1;
        if (B2 > 1) { r.S = B2; r.blen = B2; r.Pmat = c->PiR2; r.forcing = c->phi2; }
        else { r.S = B; r.blen = B; r.Pmat = PiRx; r.forcing = c->phiX; }
        r.phi = c->phiRX + (size_t)c->part_rank * phirx_chunk(c);
        if ((rc = launch_chain<2>(r, c->stream))) return rc;
    }
    return 0;   // the last rank's k_terminal wrote y_N into its second slot of phiRX (and into bndY/bndY2/yhist)
}

// y_N = L(t_N)^H lambda_N for a caller-given terminal lambda (eval_adjoint): one adjoint chain step with
// the panel of L_N as the step matrix; the result goes to every place k_terminal would write y_N
int qgdk_apply_LH(const qgdk_ctx *c)
{
    const size_t hstep = (size_t)c->Np * 2 * c->cp, panel = (size_t)c->Np * 2 * c->Np;
    ChainArgs a{};
    a.Np = c->Np; a.cp = c->cp; a.S = 1; a.Pmat = c->L + (size_t)(c->nt - 1) * panel;
    a.start = c->lam + (size_t)(c->nt - 1) * hstep; a.start_stride = 0; a.out = c->yhist + (size_t)(c->nt - 1) * hstep;
    a.forcing = c->zero_panel; a.nblocks = 1; a.blen = 1; a.ngroups = c->cp / 8;
    int rc = launch_chain<3>(a, c->stream);
    if (rc) return rc;
    const double *yN = c->yhist + (size_t)(c->nt - 1) * hstep;
    double *slot = c->phiRX + (size_t)c->part_rank * phirx_chunk(c) + hstep;
    HIPCHK(hipMemcpyAsync(slot, yN, hstep * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->bndY + (size_t)c->scan_blocks * hstep, yN, hstep * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->bndY2 + (size_t)c->scan_blocks2 * hstep, yN, hstep * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    return 0;
}

// adjoint, part 2 (after the all-gather of phiRX): y at the window end, at the super-block and block
// ends, then the history of the own blocks
int qgdk_adjoint_finish(const qgdk_ctx *c)
{
    const size_t hstep = (size_t)c->Np * 2 * c->cp, pl2 = (size_t)2 * c->Np * c->Np;
    const int B = c->scan_blocks, B2 = c->scan_blocks2, g = c->scan_g, W = c->part_world, r = c->part_rank;
    const double *PiRx = c->PiX + (size_t)B * pl2;
    int rc;
    if (chain_is_fast(c)) {
        // one launch, mirror image of the forward one: y_N is advanced over the windows of the higher ranks,
        // the super-blocks and the blocks after b, then the block writes its y history
        ChainArgs s3{};
        s3.Np = c->Np; s3.cp = c->cp; s3.S = c->nt - 1; s3.Pmat = c->Pr; s3.forcing = c->forcing; s3.out = c->yhist;
        s3.nblocks = B; s3.blen = c->scan_blen; s3.ngroups = c->cp / 8; s3.start_stride = 0;
        s3.start = (r == W - 1) ? c->yhist + (size_t)(c->nt - 1) * hstep : c->phiRX + (size_t)(W - 1) * phirx_chunk(c) + hstep;
        if (c->front) { s3.forcing = c->hforc; s3.out = c->lam; s3.start = c->lam + (size_t)(c->nt - 1) * hstep; }      // fused front: the sweep is in lambda
        int q = 0;
        if (r < W - 1) {
            s3.pre_kind[q] = 0; s3.pre_P[q] = c->RX + (size_t)(r + 1) * rx_chunk(c) + pl2; s3.pre_pm_bpr[q] = 1;
            s3.pre_pm_chunk[q] = (long long)rx_chunk(c); s3.pre_f[q] = c->phiRX + (size_t)(r + 1) * phirx_chunk(c); s3.pre_f_bpr[q] = 1;
            s3.pre_rank_count = W - 1 - r; s3.pre_start_out = c->yhist + (size_t)(c->nt - 1) * hstep; q++;
        }
        if (B2 > 1) { s3.pre_kind[q] = 1; s3.pre_P[q] = c->PiR2; s3.pre_f[q] = c->phi2; q++; }
        if (B > 1) { s3.pre_kind[q] = 2; s3.pre_P[q] = PiRx; s3.pre_f[q] = c->phiX; q++; }
        s3.npre = q; s3.pre_g = g; s3.pre_B2 = B2;
        if (suffix_on(c)) { s3.suf_P = c->SufP; s3.suf_phi = c->SufPhi; s3.suf_n = g - 2; }
        return launch_chain<3>(s3, c->stream);
    }
    if (r < W - 1) {   // y at the window end: y <- R_q^H y + phi^rank_q for q = W-1 .. r+1, from y_N
        const int nq = W - 1 - r;
        ChainArgs w{};
        w.Np = c->Np; w.cp = c->cp; w.S = nq; w.nblocks = 1; w.blen = nq; w.ngroups = c->cp / 8;
        w.Pmat = c->RX + (size_t)(r + 1) * rx_chunk(c) + pl2; w.pm_bpr = 1; w.pm_chunk = (long long)rx_chunk(c);
        w.forcing = c->phiRX + (size_t)(r + 1) * phirx_chunk(c); w.f_bpr = 1;       // slot of q is 2q: [phi | y_N] per rank
        w.start = c->phiRX + (size_t)(W - 1) * phirx_chunk(c) + hstep; w.start_stride = 0; w.out = c->wbndY;
        if ((rc = launch_chain<3>(w, c->stream))) return rc;
        // out[q'] = y before window r+1+q': y at the end of this rank's window is out[0]
        HIPCHK(hipMemcpyAsync(c->bndY + (size_t)B * hstep, c->wbndY, hstep * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->bndY2 + (size_t)B2 * hstep, c->wbndY, hstep * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->yhist + (size_t)(c->nt - 1) * hstep, c->wbndY, hstep * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    }
    if (B2 <= 1) {
        ChainArgs s2{};
        s2.Np = c->Np; s2.cp = c->cp; s2.S = B; s2.Pmat = PiRx;
        s2.start = c->bndY + (size_t)B * hstep; s2.start_stride = 0; s2.out = c->bndY;
        s2.forcing = c->phiX; s2.nblocks = 1; s2.blen = B; s2.ngroups = c->cp / 8;
        if ((rc = launch_chain<3>(s2, c->stream))) return rc;
    } else {
        ChainArgs b2{};   // y at super-block starts
        b2.Np = c->Np; b2.cp = c->cp; b2.S = B2; b2.Pmat = c->PiR2; b2.start = c->bndY2 + (size_t)B2 * hstep; b2.start_stride = 0;
        b2.out = c->bndY2; b2.forcing = c->phi2; b2.nblocks = 1; b2.blen = B2; b2.ngroups = c->cp / 8;
        if ((rc = launch_chain<3>(b2, c->stream))) return rc;
        ChainArgs c2{};   // y at every block start
        c2.Np = c->Np; c2.cp = c->cp; c2.S = B; c2.Pmat = PiRx;
        c2.start = c->bndY2 + hstep; c2.start_stride = (long long)hstep; c2.out = c->bndY; c2.forcing = c->phiX;
        c2.nblocks = B2; c2.blen = g; c2.ngroups = c->cp / 8;
        if ((rc = launch_chain<3>(c2, c->stream))) return rc;
    }
    ChainArgs s3{};
    s3.Np = c->Np; s3.cp = c->cp; s3.S = c->nt - 1; s3.Pmat = c->Pr; s3.start = c->bndY + hstep;
    s3.start_stride = (long long)hstep; s3.out = c->yhist; s3.forcing = c->forcing; s3.nblocks = B;
    s3.blen = c->scan_blen; s3.ngroups = c->cp / 8;
    return launch_chain<3>(s3, c->stream);
}

// Sensitivities of all control parameters at once (forced gradient): column group = (parameter, state
// column group).  (i) affine part of every block from zero, (ii) one chain over the block propagators of
// the forward sweep -> s at the block boundaries (the last one is s_N), (iii) with a guard: replay the
// blocks accumulating -<f_n, s_n>.
int qgdk_forced_chains(const qgdk_ctx *c)
{
    const int cpS = c->n_pcof * c->cp, B = c->scan_blocks;
    const size_t hstepS = (size_t)c->Np * 2 * cpS;
    ChainArgs f{};
    f.Np = c->Np; f.cp = cpS; f.fs_mode = 1; f.fs_m = c->m; f.fs_nops = c->n_ops; f.fs_gpc = c->cp / 8; f.fs_nt = c->g_nt ? c->g_nt : c->nt; f.fs_n0 = c->g_n0;
    f.fs_BR = c->fs_BR; f.fs_BL = c->fs_BL; f.fs_G = c->G; f.fs_goff = c->goff; f.fs_ncoef = c->ncoef; f.fs_poff = c->poff;
    ChainArgs a = f;   // (i)
    a.S = c->nt - 1; a.Pmat = c->Pc; a.phi = c->fs_phi; a.nblocks = B; a.blen = c->scan_blen; a.ngroups = cpS / 8;
    int rc = launch_chain<4>(a, c->stream);
    if (rc) return rc;
    ChainArgs s2{};    // (ii)
    s2.Np = c->Np; s2.cp = cpS; s2.S = B; s2.Pmat = c->PiX;
    s2.start = c->fs_bnd; s2.start_stride = 0; s2.out = c->fs_bnd; s2.forcing = c->fs_phi; s2.nblocks = 1; s2.blen = B;
    s2.ngroups = cpS / 8; s2.fs_gpc = c->cp / 8;
    if ((rc = launch_chain<5>(s2, c->stream))) return rc;
    if (c->have_guard) {   // (iii)
        ChainArgs g = f;
        g.S = c->nt - 1; g.Pmat = c->Pc; g.start = c->fs_bnd; g.start_stride = (long long)hstepS; g.nblocks = B; g.blen = c->scan_blen;
        g.ngroups = cpS / 8; g.fs_gf = c->forcing; g.fs_gacc = c->fs_gacc;
        if ((rc = launch_chain<5>(g, c->stream))) return rc;
    }
    return 0;
}

// forward sweep with a forcing term per step (eval_forward(...; forcing)): affine parts of the blocks,
// chain over the block propagators, history pass
int qgdk_forcing_sweep(const qgdk_ctx *c)
{
    const size_t hstep = (size_t)c->Np * 2 * c->cp;
    const int B = c->scan_blocks;
    int rc;
    ChainArgs a{};     // affine parts from zero
    a.Np = c->Np; a.cp = c->cp; a.S = c->nt - 1; a.Pmat = c->Pc; a.forcing = c->ff_Q; a.phi = c->ff_phi;
    a.nblocks = B; a.blen = c->scan_blen; a.ngroups = c->cp / 8;
    if ((rc = launch_chain<4>(a, c->stream))) return rc;
    HIPCHK(hipMemcpyAsync(c->ff_bnd, c->psi0, hstep * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    ChainArgs s2{};    // states at the block boundaries
    s2.Np = c->Np; s2.cp = c->cp; s2.S = B; s2.Pmat = c->PiX; s2.start = c->ff_bnd; s2.start_stride = 0; s2.out = c->ff_bnd;
    s2.forcing = c->ff_phi; s2.nblocks = 1; s2.blen = B; s2.ngroups = c->cp / 8;
    if ((rc = launch_chain<5>(s2, c->stream))) return rc;
    ChainArgs s3{};    // history
    s3.Np = c->Np; s3.cp = c->cp; s3.S = c->nt - 1; s3.Pmat = c->Pc; s3.start = c->ff_bnd; s3.start_stride = (long long)hstep;
    s3.out = c->hist; s3.forcing = c->ff_Q; s3.nblocks = B; s3.blen = c->scan_blen; s3.ngroups = c->cp / 8;
    return launch_chain<5>(s3, c->stream);
}

int qgdk_psi(const qgdk_ctx *c)
{
    const bool guard = c->have_guard == 2;
    hipLaunchKernelGGL(k_psi, dim3(c->nt, c->cp / 8), dim3(256), 0, c->stream, c->LinvT, c->phist, c->psi0, c->hist, c->forcing, c->hforc,
                       guard ? c->guard_diag : (const double *)nullptr, guard ? c->gpart : (double *)nullptr,
                       c->have_target ? c->target : (const double *)nullptr, c->termU, c->cp, c->nt, c->N, c->dt, c->tf, c->cost_type);
    return (int)hipGetLastError();
}

int qgdk_lambda(const qgdk_ctx *c)
{
    if (c->dense_gemm && !c->use_sparse && c->nt > 1) return qgdk_dense_lambda(c);
#define CALL_LC(N) hipLaunchKernelGGL((k_lambda_c<N>), dim3(c->cp / 8, c->nt - 1), dim3(256), 0, c->stream, c->LinvT, c->yhist, c->lam, \
                                      c->cp, c->sigma, c->nt * c->n_ops * c->m * 2, c->grad, c->grad_accumulate ? 0 : c->n_pcof)
    if (c->nt > 1) switch (c->Np) {
        case 16: CALL_LC(16); return (int)hipGetLastError();
        case 32: CALL_LC(32); return (int)hipGetLastError();
        case 48: CALL_LC(48); return (int)hipGetLastError();
        case 64: CALL_LC(64); return (int)hipGetLastError();
        default: break;
    }
#undef CALL_LC
    size_t shm = (size_t)c->Np * 16 * sizeof(double);
    hipLaunchKernelGGL(k_lambda, dim3(c->cp / 8, c->nt - 1), dim3(256), shm, c->stream, c->LinvT, c->yhist, c->lam,
                       c->Np, c->cp, c->sigma, c->nt * c->n_ops * c->m * 2, c->grad, c->grad_accumulate ? 0 : c->n_pcof);
    return (int)hipGetLastError();
}

// dynamic LDS of the chain/lambda kernels for any Np (they keep one or two panels only)
size_t qgdk_lds_needed(int Np, int m, int n_ops)
{
    size_t a = (size_t)(m + 1) * Np * 16 * sizeof(double);
    size_t b = ((size_t)2 * m * Np * 16 + (size_t)4 * n_ops * m * 2) * sizeof(double);      // (k_gradsweep: panels + per-wave scalar slots)
    return a > b ? a : b;
}


} // extern "C"
