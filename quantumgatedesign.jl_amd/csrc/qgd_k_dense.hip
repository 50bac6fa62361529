// qgd_k_dense.hip -- large N (panels do not fit in LDS): every contraction of the path as a batched
// complex GEMM on the f64 MFMA, left operands pre-arranged in "fragment order".
//
// Why a separate set of kernels.  For N > 64 the recursion kernels of qgd_k_build / qgd_k_grad assemble
// every element of A_d(t_n) = K_d - i S_d from the 2 + 2 N_op operator planes at the point of use:
// 1 + 2 N_op loads per MFMA pair, and the kernels sit at ~23 % of the MFMA peak, bound by L1 traffic.
// Here A_d(t_n) is assembled ONCE per evaluation (k_assemble_frag, 16 N^2 bytes per (n, d)) into the
// order in which a wave consumes it:
//     frag[(rb * N/4 + k4) * 64 + lane] = { Re, Im } of A(16 rb + lane%16, 4 k4 + lane/16)
// so the A operand of a 16x16x4 MFMA pair is one contiguous 1 KB global_load_dwordx4 per wave, and
// the [-Bim | Bre] right operand comes from [Bre | Bim] by a DPP row rotation instead of a second load.
// A wave owns a 32 x 32-complex-column tile (2 row blocks x 4 column groups): 6 loads per 16 MFMAs.
//
// What is computed (same mathematics as the small-N kernels; reference lines there):
//   k_level_f     D_{j+1}(t_n) = 1/(j+1) ( sum_{i=1..j} A_{j-i} D_i + A_j ),  L, R += c_{j+1} (-+dt)^{j+1} D_{j+1}
//                 (compute_derivatives! hermite.jl:56-101 on the identity, build_LHS!/build_RHS! :389-427)
//   k_derivs_f    w_{j}(t_n) = D_j(t_n) w_0(t_n), j = 1..m: m GEMMs instead of the m(m+1)/2 of the
//                 recursion, because the D_j are already there (they are what L and R were summed from)
//   k_ginit / k_gsweep_f / k_ginner_f   the O(m^2) reverse sweep and the sigma inner products of
//                 accumulate_gradient_arbitrary_fast! (eval_grad_discrete_adjoint.jl:582-800)
//
// Workgroup -> tile mapping is XCD-aware: consecutive workgroup ids go round-robin to the 8 XCDs, so
// id%8 selects the XCD and all tiles of one time point (which share A_d(t_n) and the panels of t_n)
// are given ids that land on the same XCD, i.e. in the same L2.
//
// Round 3:
//   * three real products per complex multiply (cgemm3_tile, cgemm3_tile_planes, cgemm3_tile_panelH, outer_tile3, outer_frag3;
//     kernels *_f3, k_propagator3, k_chain_step; QGD_PATHS=dense_4m keeps the four-product kernels beside them),
//   * the gradient scalars from outer products (k_youter, k_yinit, k_gouter, k_ginner_m, k_ginner_d) with fixed-order sums,
//   * the inverse as block Gauss-Jordan over 64-column blocks (k_binv_row, k_binv_rest, k_binv_planes + k_inverse_diag),
//   * one chain step as a full-chip GEMM (k_chain_step) for the sequential top-level chains of the scan.
#include "qgd_kernels_common.h"
#include <algorithm>

typedef double d2 __attribute__((ext_vector_type(2)));

// Tile shapes <RB, NG>: RB 16-row blocks x NG 16-wide column groups per wave (4 waves stack their row blocks).
// <2,4> when there are at least 3 column groups; <4,2> and <4,1> for 2 and 1 groups (few initial conditions):
// the same 64 or 32 accumulator registers and nearly the same loads per MFMA, no MFMAs spent on padding columns.

// [-Bim | Bre] from [Bre | Bim]: rotate the 16-lane row by 8 and negate lanes 0..7
__device__ __forceinline__ double swap8_signed(double b1, int sign_hi)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(b1), 0x128, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(b1), 0x128, 0xF, 0xF, false);
    return __hiloint2double(hi ^ sign_hi, lo);
}

// position of element (row, k) in fragment order
__device__ __forceinline__ size_t frag_index(int Np, int row, int k)
{
    return ((size_t)(row >> 4) * (Np >> 2) + (k >> 2)) * 64 + (size_t)((k & 3) * 16 + (row & 15));
}

template <int DN_RB, int DN_NG>
struct DenseTile {
    int n, sub, rb[DN_RB], g[DN_NG];     // time point, sub-index (level / source), row blocks and groups (-1 = outside)
    int slot, ntile;                     // index of the workgroup among the T = ntile * nsub workgroups of its time point (ntile per sub-index)
    int lane, c16, kk, sign_hi;
};

// grid = 8 * T * ceil(nt / 8) workgroups, T = workgroups per time point.  A WAVE owns one item = (group of DN_RB row blocks,
// tile of DN_NG column groups); the items of one sub-index are dealt to the workgroups four at a time, row-block groups
// fastest (the waves of a workgroup mostly share their right operand through the L1, and share nothing else).  Round 3,
// last change: before, a workgroup was 4 * DN_RB consecutive row blocks of ONE column tile, and Np = 144 (9 row blocks)
// paid for two 128-row tiles; now its 5 row-block pairs x ctiles items fill ceil(5 ctiles / 4) workgroups.
template <int DN_RB, int DN_NG>
__device__ __forceinline__ bool dense_tile(DenseTile<DN_RB, DN_NG> &t, int nrb, int ngroups, int nsub, int nt, bool sub_per_wave = false)
{
    // sub_per_wave: the sub-index belongs to the ITEM, not to the workgroup (kernels whose waves share nothing that depends on
    // it): nsub * per_sub items are dealt four to a workgroup -- with few columns per_sub is 1-3 and half the waves idled
    const int nrp = (nrb + DN_RB - 1) / DN_RB, ctiles = (ngroups + DN_NG - 1) / DN_NG;
    const int per_sub = nrp * ctiles, total = sub_per_wave ? per_sub * nsub : per_sub, wgs = (total + 3) >> 2;
    const int T = sub_per_wave ? wgs : wgs * nsub;
    const int xcd = blockIdx.x & 7, s = blockIdx.x >> 3;
    t.n = (s / T) * 8 + xcd;
    int r = s % T;
    // The last group holds only nt % 8 time points: with "time point = XCD" its tiles would all land on the first nt % 8
    // XCDs while the others idle (config 5: 201 time points = 25 groups + 1 point -- XCD 0 did 26 units of work, the
    // other seven 25: 3.4 % of every per-time-point kernel).  The tiles of the tail are dealt over all eight XCDs instead.
    const int full = nt & ~7, rem = nt - full;
    if (rem && t.n >= full) {
        const int q = r * 8 + xcd;                     // flat slot of the tail group, 8 T of them
        if (q >= rem * T) { t.n = nt; return false; }
        t.n = full + q / T;
        r = q % T;
    }
    if (t.n >= nt) return false;
    t.slot = r; t.ntile = wgs;
    t.sub = 0;
    if (!sub_per_wave) { t.sub = r / wgs; r %= wgs; }
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    t.lane = threadIdx.x & 63; t.c16 = t.lane & 15; t.kk = t.lane >> 4;
    t.sign_hi = (t.c16 < 8) ? (int)0x80000000 : 0;
    int item = r * 4 + wave;
    const bool on = item < total;
    if (!on) item = 0;
    if (sub_per_wave) { t.sub = item / per_sub; item %= per_sub; }
    const int rp = item % nrp, ct = item / nrp;
    #pragma unroll
    for (int i = 0; i < DN_RB; i++) { const int rb = rp * DN_RB + i; t.rb[i] = (on && rb < nrb) ? rb : -1; }
    #pragma unroll
    for (int i = 0; i < DN_NG; i++) { const int g = ct * DN_NG + i; t.g[i] = g < ngroups ? g : -1; }
    return t.rb[0] >= 0;
}

static inline int dense_grid(int DN_RB, int DN_NG, int nrb, int ngroups, int nsub, int nt, bool sub_per_wave = false)
{
    const int nrp = (nrb + DN_RB - 1) / DN_RB, ctiles = (ngroups + DN_NG - 1) / DN_NG, per_sub = nrp * ctiles;
    const int T = sub_per_wave ? (per_sub * nsub + 3) / 4 : ((per_sub + 3) / 4) * nsub;
    return 8 * T * ((nt + 7) / 8);
}

// acc += A * B for the wave's tile.  A: fragment order (complex); B: panel with row stride ldb.
// Row blocks / groups outside the matrix are clamped to the first one (computed, never stored).
template <int DN_RB, int DN_NG>
__device__ __forceinline__ void cgemm_tile(d4 (&acc)[DN_RB][DN_NG], const DenseTile<DN_RB, DN_NG> &t, const d2 *__restrict__ A,
                                           const double *__restrict__ B, size_t ldb, int Np)
{
    const d2 *ap[DN_RB];
    const double *bp[DN_NG];
    #pragma unroll
    for (int r = 0; r < DN_RB; r++) ap[r] = A + (size_t)(t.rb[r] >= 0 ? t.rb[r] : t.rb[0]) * (Np >> 2) * 64 + t.lane;
    #pragma unroll
    for (int g = 0; g < DN_NG; g++) bp[g] = B + (size_t)t.kk * ldb + (size_t)(t.g[g] >= 0 ? t.g[g] : t.g[0]) * 16 + t.c16;
    const int nk4 = Np >> 2;
    // software pipeline: the operands of step k4+1 are in flight while the 16 MFMAs of step k4 issue
    d2 a[DN_RB], an[DN_RB];
    double b[DN_NG], bn[DN_NG];
    #pragma unroll
    for (int r = 0; r < DN_RB; r++) a[r] = ap[r][0];
    #pragma unroll
    for (int g = 0; g < DN_NG; g++) b[g] = bp[g][0];
    for (int k4 = 0; k4 < nk4; k4++) {
        const int kn = (k4 + 1 < nk4) ? k4 + 1 : k4;
        #pragma unroll
        for (int r = 0; r < DN_RB; r++) an[r] = ap[r][(size_t)kn * 64];
        #pragma unroll
        for (int g = 0; g < DN_NG; g++) bn[g] = bp[g][(size_t)kn * 4 * ldb];
        #pragma unroll
        for (int g = 0; g < DN_NG; g++) {
            const double b2 = swap8_signed(b[g], t.sign_hi);
            #pragma unroll
            for (int r = 0; r < DN_RB; r++) {
                acc[r][g] = MFMA(a[r].x, b[g], acc[r][g]);
                acc[r][g] = MFMA(a[r].y, b2, acc[r][g]);
            }
        }
        #pragma unroll
        for (int r = 0; r < DN_RB; r++) a[r] = an[r];
        #pragma unroll
        for (int g = 0; g < DN_NG; g++) b[g] = bn[g];
    }
}

// acc += A^H * B with A given as two column-major planes of A^T (re, im): element (A^H)(row, k) = conj(A^T'(...)),
// i.e. planes indexed [row + Np*k] hold A(k, row).  The conjugation is a sign on the rotated right operand.
template <int DN_RB, int DN_NG>
__device__ __forceinline__ void cgemm_tile_planes_conj(d4 (&acc)[DN_RB][DN_NG], const DenseTile<DN_RB, DN_NG> &t,
                                                       const double *__restrict__ Are, const double *__restrict__ Aim,
                                                       const double *__restrict__ B, size_t ldb, int Np)
{
    const double *apr[DN_RB], *api[DN_RB];
    const double *bp[DN_NG];
    #pragma unroll
    for (int r = 0; r < DN_RB; r++) {
        const size_t o = (size_t)(t.rb[r] >= 0 ? t.rb[r] : t.rb[0]) * 16 + t.c16 + (size_t)Np * t.kk;
        apr[r] = Are + o; api[r] = Aim + o;
    }
    #pragma unroll
    for (int g = 0; g < DN_NG; g++) bp[g] = B + (size_t)t.kk * ldb + (size_t)(t.g[g] >= 0 ? t.g[g] : t.g[0]) * 16 + t.c16;
    const int nk4 = Np >> 2, sgn = t.sign_hi ^ (int)0x80000000;
    double ar[DN_RB], ai[DN_RB], nr[DN_RB], ni[DN_RB], b[DN_NG], bn[DN_NG];
    #pragma unroll
    for (int r = 0; r < DN_RB; r++) { ar[r] = apr[r][0]; ai[r] = api[r][0]; }
    #pragma unroll
    for (int g = 0; g < DN_NG; g++) b[g] = bp[g][0];
    for (int k4 = 0; k4 < nk4; k4++) {
        const int kn = (k4 + 1 < nk4) ? k4 + 1 : k4;
        #pragma unroll
        for (int r = 0; r < DN_RB; r++) { nr[r] = apr[r][(size_t)kn * 4 * Np]; ni[r] = api[r][(size_t)kn * 4 * Np]; }
        #pragma unroll
        for (int g = 0; g < DN_NG; g++) bn[g] = bp[g][(size_t)kn * 4 * ldb];
        #pragma unroll
        for (int g = 0; g < DN_NG; g++) {
            const double b2 = swap8_signed(b[g], sgn);
            #pragma unroll
            for (int r = 0; r < DN_RB; r++) {
                acc[r][g] = MFMA(ar[r], b[g], acc[r][g]);
                acc[r][g] = MFMA(ai[r], b2, acc[r][g]);
            }
        }
        #pragma unroll
        for (int r = 0; r < DN_RB; r++) { ar[r] = nr[r]; ai[r] = ni[r]; }
        #pragma unroll
        for (int g = 0; g < DN_NG; g++) b[g] = bn[g];
    }
}

// U += S * B, V += K * B with real left operands packed as {S, K} pairs in fragment order
template <int DN_RB, int DN_NG>
__device__ __forceinline__ void rgemm2_tile(d4 (&U)[DN_RB][DN_NG], d4 (&V)[DN_RB][DN_NG], const DenseTile<DN_RB, DN_NG> &t,
                                            const d2 *__restrict__ A, const double *__restrict__ B, size_t ldb, int Np)
{
    const d2 *ap[DN_RB];
    const double *bp[DN_NG];
    #pragma unroll
    for (int r = 0; r < DN_RB; r++) ap[r] = A + (size_t)(t.rb[r] >= 0 ? t.rb[r] : t.rb[0]) * (Np >> 2) * 64 + t.lane;
    #pragma unroll
    for (int g = 0; g < DN_NG; g++) bp[g] = B + (size_t)t.kk * ldb + (size_t)(t.g[g] >= 0 ? t.g[g] : t.g[0]) * 16 + t.c16;
    const int nk4 = Np >> 2;
    d2 a[DN_RB], an[DN_RB];
    double b[DN_NG], bn[DN_NG];
    #pragma unroll
    for (int r = 0; r < DN_RB; r++) a[r] = ap[r][0];
    #pragma unroll
    for (int g = 0; g < DN_NG; g++) b[g] = bp[g][0];
    for (int k4 = 0; k4 < nk4; k4++) {
        const int kn = (k4 + 1 < nk4) ? k4 + 1 : k4;
        #pragma unroll
        for (int r = 0; r < DN_RB; r++) an[r] = ap[r][(size_t)kn * 64];
        #pragma unroll
        for (int g = 0; g < DN_NG; g++) bn[g] = bp[g][(size_t)kn * 4 * ldb];
        #pragma unroll
        for (int g = 0; g < DN_NG; g++)
            #pragma unroll
            for (int r = 0; r < DN_RB; r++) {
                U[r][g] = MFMA(a[r].x, b[g], U[r][g]);
                V[r][g] = MFMA(a[r].y, b[g], V[r][g]);
            }
        #pragma unroll
        for (int r = 0; r < DN_RB; r++) a[r] = an[r];
        #pragma unroll
        for (int g = 0; g < DN_NG; g++) b[g] = bn[g];
    }
}

#define ZERO_ACC(acc) _Pragma("unroll") for (int r_ = 0; r_ < DN_RB; r_++) _Pragma("unroll") for (int g_ = 0; g_ < DN_NG; g_++) acc[r_][g_] = (d4){0, 0, 0, 0}

// ---------------------------------------------------------------------------
// The same complex product with THREE real products per complex multiply instead of four ("3M"):
//     P1 = Are Bre,  P2 = Aim Bim,  P3 = (Are + Aim)(Bre + Bim);   Re C = P1 - P2,  Im C = P3 - P1 - P2.
// An MFMA wants 16 real columns of ONE kind, so two column groups are taken together: lanes c16 < 8 work on the 8
// complex columns of group g[2p], lanes c16 >= 8 on those of group g[2p+1] -- the lane's address picks the real half
// (+0) or the imaginary half (+8) of ITS group, no cross-lane movement at all.  Per k-step of a <2, 4> tile: 12 MFMAs
// instead of 16, 6 loads, 4 f64 adds (the two operand sums); three accumulators per 16 x 16 block instead of two (96
// instead of 64 registers).  The result of lane (c16, kk), element e, is the complex entry (row kk + 4e of the block,
// column 8 g + c16 % 8) with both parts in the same lane.  Rounding: the imaginary part is exact to eps * |A||B|
// (norm-wise) instead of component-wise -- 1e-16-level on this path's matrices (tests/test_gpu_large_n.py).
// ---------------------------------------------------------------------------
template <int DN_RB, int DN_NG>
__device__ __forceinline__ int lane_group(const DenseTile<DN_RB, DN_NG> &t, int p)     // -1: the lane's columns are outside
{
    return (t.c16 < 8) ? t.g[2 * p] : t.g[2 * p + 1];
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ d2 buffer_load_d2(__amdgpu_buffer_rsrc_t r, int voff, int soff)
{
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return (d2){__hiloint2double((int)v.y, (int)v.x), __hiloint2double((int)v.w, (int)v.z)};
}

// The k-loop carries no vector-ALU instruction but the four operand sums: buffer-addressed loads (per-lane offset fixed,
// the k-step offset in an SGPR) and two register sets that swap roles by unrolling -- beside twelve f64 MFMAs per k-step
// every other vector instruction costs about a sixth of an MFMA slot, and the pointer updates and
// register copies of the plain loop were 12 of them (MFMA pipe busy 0.73 -> see DESIGN.md section 4b).
template <int DN_RB, int DN_NG>
__device__ __forceinline__ void cgemm3_tile(d4 (&p1)[DN_RB][DN_NG / 2], d4 (&p2)[DN_RB][DN_NG / 2], d4 (&p3)[DN_RB][DN_NG / 2],
                                            const DenseTile<DN_RB, DN_NG> &t, const d2 *__restrict__ A,
                                            const double *__restrict__ B, size_t ldb, int Np)
{
    constexpr int NP = DN_NG / 2;
    const __amdgpu_buffer_rsrc_t ra = buffer_of(reinterpret_cast<const double *>(A)), rb_ = buffer_of(B);
    int av[DN_RB], bv[NP];
    #pragma unroll
    for (int r = 0; r < DN_RB; r++) av[r] = ((t.rb[r] >= 0 ? t.rb[r] : t.rb[0]) * (Np >> 2) * 64 + t.lane) * 16;
    #pragma unroll
    for (int p = 0; p < NP; p++) {
        int g = lane_group(t, p);
        if (g < 0) g = t.g[0];
        bv[p] = (t.kk * (int)ldb + g * 16 + (t.c16 & 7)) * 8;
    }
    const int nk4 = Np >> 2, bstep = 4 * (int)ldb * 8;      // (Np is a multiple of 16: nk4 is even)
    d2 a0[DN_RB], a1[DN_RB];
    double br0[NP], bi0[NP], br1[NP], bi1[NP];
#define C3_LOAD(a, br, bi, k) do {                                                                            \
        _Pragma("unroll") for (int r = 0; r < DN_RB; r++) a[r] = buffer_load_d2(ra, av[r], (k) * 1024);          \
        _Pragma("unroll") for (int p = 0; p < NP; p++) { br[p] = buffer_load_f64(rb_, bv[p], (k) * bstep);       \
                                                         bi[p] = buffer_load_f64(rb_, bv[p] + 64, (k) * bstep); } } while (0)
#define C3_MFMA(a, br, bi) do {                                                                               \
        double as[DN_RB], bs[NP];                                                                             \
        _Pragma("unroll") for (int r = 0; r < DN_RB; r++) as[r] = a[r].x + a[r].y;                              \
        _Pragma("unroll") for (int p = 0; p < NP; p++) bs[p] = br[p] + bi[p];                                   \
        _Pragma("unroll") for (int p = 0; p < NP; p++)                                                         \
            _Pragma("unroll") for (int r = 0; r < DN_RB; r++) {                                                \
                p1[r][p] = MFMA(a[r].x, br[p], p1[r][p]);                                                      \
                p2[r][p] = MFMA(a[r].y, bi[p], p2[r][p]);                                                      \
                p3[r][p] = MFMA(as[r], bs[p], p3[r][p]);                                                       \
            } } while (0)
    C3_LOAD(a0, br0, bi0, 0);
    for (int k4 = 0; k4 < nk4; k4 += 2) {
        C3_LOAD(a1, br1, bi1, k4 + 1);
        C3_MFMA(a0, br0, bi0);
        const int kn = (k4 + 2 < nk4) ? k4 + 2 : k4;
        C3_LOAD(a0, br0, bi0, kn);
        C3_MFMA(a1, br1, bi1);
    }
#undef C3_LOAD
#undef C3_MFMA
}

// ... with the left operand given as two column-major planes, element (row, k) at [row + Np k] (the inverse as the blocked
// elimination writes it); CONJ: the operand is conj(planes), i.e. s = -1 in
//     P1 = ar Br,  P2 = ai Bi,  P3 = (ar + s ai)(Br + Bi);   Re = P1 - s P2,  Im = P3 - P1 - s P2.
template <bool CONJ, int DN_RB, int DN_NG>
__device__ __forceinline__ void cgemm3_tile_planes(d4 (&p1)[DN_RB][DN_NG / 2], d4 (&p2)[DN_RB][DN_NG / 2], d4 (&p3)[DN_RB][DN_NG / 2],
                                                   const DenseTile<DN_RB, DN_NG> &t, const double *__restrict__ Are,
                                                   const double *__restrict__ Aim, const double *__restrict__ B, size_t ldb, int Np,
                                                   int K = -1)        // Np: leading dimension of the planes; K: contraction length (default Np)
{
    constexpr int NP = DN_NG / 2;
    const __amdgpu_buffer_rsrc_t rr = buffer_of(Are), ri = buffer_of(Aim), rb_ = buffer_of(B);
    if (K < 0) K = Np;
    int av[DN_RB], bv[NP];
    #pragma unroll
    for (int r = 0; r < DN_RB; r++) av[r] = ((t.rb[r] >= 0 ? t.rb[r] : t.rb[0]) * 16 + t.c16 + Np * t.kk) * 8;
    #pragma unroll
    for (int p = 0; p < NP; p++) {
        int g = lane_group(t, p);
        if (g < 0) g = t.g[0];
        bv[p] = (t.kk * (int)ldb + g * 16 + (t.c16 & 7)) * 8;
    }
    const int nk4 = K >> 2, astep = 4 * Np * 8, bstep = 4 * (int)ldb * 8;      // (K is a multiple of 16: nk4 is even)
    double ar0[DN_RB], ai0[DN_RB], ar1[DN_RB], ai1[DN_RB], br0[NP], bi0[NP], br1[NP], bi1[NP];
#define CP_LOAD(ar, ai, br, bi, k) do {                                                                        \
        _Pragma("unroll") for (int r = 0; r < DN_RB; r++) { ar[r] = buffer_load_f64(rr, av[r], (k) * astep);     \
                                                            ai[r] = buffer_load_f64(ri, av[r], (k) * astep); }   \
        _Pragma("unroll") for (int p = 0; p < NP; p++) { br[p] = buffer_load_f64(rb_, bv[p], (k) * bstep);       \
                                                         bi[p] = buffer_load_f64(rb_, bv[p] + 64, (k) * bstep); } } while (0)
#define CP_MFMA(ar, ai, br, bi) do {                                                                           \
        double as[DN_RB], bs[NP];                                                                             \
        _Pragma("unroll") for (int r = 0; r < DN_RB; r++) as[r] = CONJ ? ar[r] - ai[r] : ar[r] + ai[r];          \
        _Pragma("unroll") for (int p = 0; p < NP; p++) bs[p] = br[p] + bi[p];                                   \
        _Pragma("unroll") for (int p = 0; p < NP; p++)                                                         \
            _Pragma("unroll") for (int r = 0; r < DN_RB; r++) {                                                \
                p1[r][p] = MFMA(ar[r], br[p], p1[r][p]);                                                       \
                p2[r][p] = MFMA(ai[r], bi[p], p2[r][p]);                                                       \
                p3[r][p] = MFMA(as[r], bs[p], p3[r][p]);                                                       \
            } } while (0)
    CP_LOAD(ar0, ai0, br0, bi0, 0);
    for (int k4 = 0; k4 < nk4; k4 += 2) {
        CP_LOAD(ar1, ai1, br1, bi1, k4 + 1);
        CP_MFMA(ar0, ai0, br0, bi0);
        const int kn = (k4 + 2 < nk4) ? k4 + 2 : k4;
        CP_LOAD(ar0, ai0, br0, bi0, kn);
        CP_MFMA(ar1, ai1, br1, bi1);
    }
#undef CP_LOAD
#undef CP_MFMA
}

// ... with the left operand P^H taken from P's PANEL layout (row k of P holds [8 re | 8 im] groups over its columns):
// element (row, k) of the operand = conj(P(k, row)), the eight lanes c16 & 7 read eight consecutive doubles (chain_a_raw<ADJ>
// of qgd_k_chain.hip).  s = -1:  P1 = ar Br, P2 = ai Bi, P3 = (ar - ai)(Br + Bi);  Re = P1 + P2,  Im = P3 - P1 + P2.
template <int DN_RB, int DN_NG>
__device__ __forceinline__ void cgemm3_tile_panelH(d4 (&p1)[DN_RB][DN_NG / 2], d4 (&p2)[DN_RB][DN_NG / 2], d4 (&p3)[DN_RB][DN_NG / 2],
                                                   const DenseTile<DN_RB, DN_NG> &t, const double *__restrict__ P,
                                                   const double *__restrict__ B, size_t ldb, int Np, int K = -1)
{
    constexpr int NP = DN_NG / 2;
    const __amdgpu_buffer_rsrc_t rp = buffer_of(P), rb_ = buffer_of(B);
    if (K < 0) K = Np;
    int av[DN_RB], bv[NP];
    #pragma unroll
    for (int r = 0; r < DN_RB; r++) {
        const int arow = (t.rb[r] >= 0 ? t.rb[r] : t.rb[0]) * 16 + t.c16;
        av[r] = (t.kk * 2 * Np + (arow >> 3) * 16 + (arow & 7)) * 8;
    }
    #pragma unroll
    for (int p = 0; p < NP; p++) {
        int g = lane_group(t, p);
        if (g < 0) g = t.g[0];
        bv[p] = (t.kk * (int)ldb + g * 16 + (t.c16 & 7)) * 8;
    }
    const int nk4 = K >> 2, astep = 4 * 2 * Np * 8, bstep = 4 * (int)ldb * 8;
    double ar0[DN_RB], ai0[DN_RB], ar1[DN_RB], ai1[DN_RB], br0[NP], bi0[NP], br1[NP], bi1[NP];
#define CH_LOAD(ar, ai, br, bi, k) do {                                                                        \
        _Pragma("unroll") for (int r = 0; r < DN_RB; r++) { ar[r] = buffer_load_f64(rp, av[r], (k) * astep);     \
                                                            ai[r] = buffer_load_f64(rp, av[r] + 64, (k) * astep); } \
        _Pragma("unroll") for (int p = 0; p < NP; p++) { br[p] = buffer_load_f64(rb_, bv[p], (k) * bstep);       \
                                                         bi[p] = buffer_load_f64(rb_, bv[p] + 64, (k) * bstep); } } while (0)
#define CH_MFMA(ar, ai, br, bi) do {                                                                           \
        double as[DN_RB], bs[NP];                                                                             \
        _Pragma("unroll") for (int r = 0; r < DN_RB; r++) as[r] = ar[r] - ai[r];                                \
        _Pragma("unroll") for (int p = 0; p < NP; p++) bs[p] = br[p] + bi[p];                                   \
        _Pragma("unroll") for (int p = 0; p < NP; p++)                                                         \
            _Pragma("unroll") for (int r = 0; r < DN_RB; r++) {                                                \
                p1[r][p] = MFMA(ar[r], br[p], p1[r][p]);                                                       \
                p2[r][p] = MFMA(ai[r], bi[p], p2[r][p]);                                                       \
                p3[r][p] = MFMA(as[r], bs[p], p3[r][p]);                                                       \
            } } while (0)
    CH_LOAD(ar0, ai0, br0, bi0, 0);
    for (int k4 = 0; k4 < nk4; k4 += 2) {
        CH_LOAD(ar1, ai1, br1, bi1, k4 + 1);
        CH_MFMA(ar0, ai0, br0, bi0);
        const int kn = (k4 + 2 < nk4) ? k4 + 2 : k4;
        CH_LOAD(ar0, ai0, br0, bi0, kn);
        CH_MFMA(ar1, ai1, br1, bi1);
    }
#undef CH_LOAD
#undef CH_MFMA
}

#define ZERO_ACC3(acc) _Pragma("unroll") for (int r_ = 0; r_ < DN_RB; r_++) _Pragma("unroll") for (int g_ = 0; g_ < DN_NG / 2; g_++) acc[r_][g_] = (d4){0, 0, 0, 0}
// 3M switch of the N > 64 kernels (QGD_PATHS=dense_4m keeps the four-product tiles, which some tile shapes take anyway:
// tests and A/B timing)
static bool dense_3m()
{
    return qgd_path("dense_4m") == nullptr;          // (read per call: tests switch it inside one process; noise beside these launches)
}

// ---------------------------------------------------------------------------
// A_d(t_n), d = 0..m-1, in fragment order.  grid (Np^2/256, m, nt)
// ---------------------------------------------------------------------------
template <int NOPS>
__global__ __launch_bounds__(256) void k_assemble_frag(const double *__restrict__ ops, const double *__restrict__ tab,
                                                       d2 *__restrict__ Afrag, int Np, int n_ops, int m)
{
    const int n = blockIdx.z, d = blockIdx.y;
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int lane = e & 63;
    const size_t f = e >> 6;
    const int k4 = (int)(f % (Np >> 2)), rb = (int)(f / (Np >> 2));
    OpCoef cf;
    load_coef(cf, tab, n, d, m, n_ops);
    double are, aim;
    assembled_a<NOPS>(ops, Np, n_ops, cf, rb * 16 + (lane & 15), k4 * 4 + (lane >> 4), are, aim);
    Afrag[((size_t)n * m + d) * Np * Np + e] = (d2){are, aim};
}

// { S_o, K_o } pairs of the control operators in fragment order (once per problem).  grid (Np^2/256, n_ops)
__global__ __launch_bounds__(256) void k_operator_frag(const double *__restrict__ ops, d2 *__restrict__ OpFrag, int Np)
{
    const int o = blockIdx.y;
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int lane = e & 63;
    const size_t f = e >> 6;
    const int k4 = (int)(f % (Np >> 2)), rb = (int)(f / (Np >> 2));
    const size_t pl = (size_t)Np * Np;
    const size_t src = (size_t)(rb * 16 + (lane & 15)) + (size_t)Np * (k4 * 4 + (lane >> 4));
    OpFrag[(size_t)o * pl + e] = (d2){ops[(size_t)(3 + 2 * o) * pl + src], ops[(size_t)(2 + 2 * o) * pl + src]};
}

// ---------------------------------------------------------------------------
// one level of the recursion on the identity; D: [nt][m] panels [Np][2Np], Dfrag: the same in fragment order
// ---------------------------------------------------------------------------
template <int DN_RB, int DN_NG>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_level_f(const d2 *__restrict__ Afrag, double *D,
                                                 double *__restrict__ Dfrag, double *__restrict__ L,
                                                 double *__restrict__ R, const double *__restrict__ cw, int Np, int m, int nt, int j,
                                                 double cL, double cR)
{
    DenseTile<DN_RB, DN_NG> t;
    if (!dense_tile(t, Np >> 4, Np >> 3, 1, nt)) return;
    const int PW = 2 * Np;
    const size_t panel = (size_t)Np * PW, fr = (size_t)Np * Np;
    double *Dn = D + (size_t)t.n * m * panel;
    const d2 *An = Afrag + (size_t)t.n * m * fr;
    d4 acc[DN_RB][DN_NG];
    ZERO_ACC(acc);
    for (int i = 1; i <= j; i++) cgemm_tile(acc, t, An + (size_t)(j - i) * fr, Dn + (size_t)(i - 1) * panel, PW, Np);
    const double inv = 1.0 / (double)(j + 1);
    const double *Aj = reinterpret_cast<const double *>(An + (size_t)j * fr);
    double *Dout = Dn + (size_t)j * panel, *Fout = Dfrag + ((size_t)t.n * m + j) * 2 * fr;
    double *Ln = L + (size_t)t.n * panel, *Rn = R + (size_t)t.n * panel;
    const int is_im = t.c16 >> 3;
    #pragma unroll
    for (int r = 0; r < DN_RB; r++) {
        if (t.rb[r] < 0) continue;
        #pragma unroll
        for (int g = 0; g < DN_NG; g++) {
            if (t.g[g] < 0) continue;
            const int ccol = t.g[g] * 8 + (t.c16 & 7);
            #pragma unroll
            for (int e = 0; e < 4; e++) {
                const int row = t.rb[r] * 16 + t.kk + 4 * e;
                const size_t fi = 2 * frag_index(Np, row, ccol) + is_im;
                const double val = (acc[r][g][e] + Aj[fi]) * inv;
                const size_t o = (size_t)row * PW + t.g[g] * 16 + t.c16;
                Dout[o] = val;
                Fout[fi] = val;
                if (j == m - 1) {       // L and R once, from the stored D_1..D_{m-1} (not a read-modify-write per level)
                    const double id = (!is_im && row == ccol) ? 1.0 : 0.0;
                    double lv = id + cL * val, rv = id + cR * val;
                    for (int q = 0; q < j; q++) {
                        const double dq = Dn[(size_t)q * panel + o];
                        lv += cw[2 * (q + 1) + 1] * dq;
                        rv += cw[2 * (q + 1)] * dq;
                    }
                    Ln[o] = lv;
                    Rn[o] = rv;
                }
            }
        }
    }
}

// the same level with the three-product tiles (cgemm3_tile)
template <int DN_RB, int DN_NG>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_level_f3(const d2 *__restrict__ Afrag, double *D,
                                                 double *__restrict__ Dfrag, double *__restrict__ L,
                                                 double *__restrict__ R, const double *__restrict__ cw, int Np, int m, int nt, int j,
                                                 double cL, double cR)
{
    DenseTile<DN_RB, DN_NG> t;
    if (!dense_tile(t, Np >> 4, Np >> 3, 1, nt)) return;
    const int PW = 2 * Np;
    const size_t panel = (size_t)Np * PW, fr = (size_t)Np * Np;
    double *Dn = D + (size_t)t.n * m * panel;
    const d2 *An = Afrag + (size_t)t.n * m * fr;
    d4 p1[DN_RB][DN_NG / 2], p2[DN_RB][DN_NG / 2], p3[DN_RB][DN_NG / 2];
    ZERO_ACC3(p1); ZERO_ACC3(p2); ZERO_ACC3(p3);
    for (int i = 1; i <= j; i++) cgemm3_tile(p1, p2, p3, t, An + (size_t)(j - i) * fr, Dn + (size_t)(i - 1) * panel, PW, Np);
    const double inv = 1.0 / (double)(j + 1);
    const d2 *Aj = An + (size_t)j * fr;
    double *Dout = Dn + (size_t)j * panel;
    d2 *Fout = reinterpret_cast<d2 *>(Dfrag) + ((size_t)t.n * m + j) * fr;
    double *Ln = L + (size_t)t.n * panel, *Rn = R + (size_t)t.n * panel;
    #pragma unroll
    for (int r = 0; r < DN_RB; r++) {
        if (t.rb[r] < 0) continue;
        #pragma unroll
        for (int p = 0; p < DN_NG / 2; p++) {
            const int g = lane_group(t, p);
            if (g < 0) continue;
            const int ccol = g * 8 + (t.c16 & 7);
            #pragma unroll
            for (int e = 0; e < 4; e++) {
                const int row = t.rb[r] * 16 + t.kk + 4 * e;
                const size_t fi = frag_index(Np, row, ccol);
                const d2 aj = Aj[fi];
                const double s12 = p1[r][p][e] + p2[r][p][e];
                const double vre = (p1[r][p][e] - p2[r][p][e] + aj.x) * inv, vim = (p3[r][p][e] - s12 + aj.y) * inv;
                const size_t o = (size_t)row * PW + g * 16 + (t.c16 & 7);
                Dout[o] = vre; Dout[o + 8] = vim;
                Fout[fi] = (d2){vre, vim};
                if (j == m - 1) {
                    const double id = (row == ccol) ? 1.0 : 0.0;
                    double lre = id + cL * vre, rre = id + cR * vre, lim = cL * vim, rim = cR * vim;
                    for (int q = 0; q < j; q++) {
                        const double dre = Dn[(size_t)q * panel + o], dim = Dn[(size_t)q * panel + o + 8];
                        const double wl = cw[2 * (q + 1) + 1], wr = cw[2 * (q + 1)];
                        lre += wl * dre; lim += wl * dim;
                        rre += wr * dre; rim += wr * dim;
                    }
                    Ln[o] = lre; Ln[o + 8] = lim;
                    Rn[o] = rre; Rn[o + 8] = rim;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// w_{j+1}(t_n) = D_{j+1}(t_n) w_0(t_n), j = 0..m-1 (sub-index).  dpsi: [nt][m] panels [Np][2cp]
// ---------------------------------------------------------------------------
template <int DN_RB, int DN_NG>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_derivs_f(const d2 *__restrict__ Dfrag, const double *__restrict__ hist,
                                                  double *__restrict__ dpsi, int Np, int cp, int m, int nt)
{
    DenseTile<DN_RB, DN_NG> t;
    if (!dense_tile(t, Np >> 4, cp >> 3, m, nt, true)) return;
    const int PWc = 2 * cp;
    const size_t hstep = (size_t)Np * PWc, fr = (size_t)Np * Np;
    d4 acc[DN_RB][DN_NG];
    ZERO_ACC(acc);
    cgemm_tile(acc, t, Dfrag + ((size_t)t.n * m + t.sub) * fr, hist + (size_t)t.n * hstep, PWc, Np);
    double *out = dpsi + ((size_t)t.n * m + t.sub) * hstep;
    #pragma unroll
    for (int r = 0; r < DN_RB; r++) {
        if (t.rb[r] < 0) continue;
        #pragma unroll
        for (int g = 0; g < DN_NG; g++) {
            if (t.g[g] < 0) continue;
            #pragma unroll
            for (int e = 0; e < 4; e++)
                out[(size_t)(t.rb[r] * 16 + t.kk + 4 * e) * PWc + t.g[g] * 16 + t.c16] = acc[r][g][e];
        }
    }
}

template <int DN_RB, int DN_NG>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_derivs_f3(const d2 *__restrict__ Dfrag, const double *__restrict__ hist,
                                                  double *__restrict__ dpsi, int Np, int cp, int m, int nt)
{
    DenseTile<DN_RB, DN_NG> t;
    if (!dense_tile(t, Np >> 4, cp >> 3, m, nt, true)) return;
    const int PWc = 2 * cp;
    const size_t hstep = (size_t)Np * PWc, fr = (size_t)Np * Np;
    d4 p1[DN_RB][DN_NG / 2], p2[DN_RB][DN_NG / 2], p3[DN_RB][DN_NG / 2];
    ZERO_ACC3(p1); ZERO_ACC3(p2); ZERO_ACC3(p3);
    cgemm3_tile(p1, p2, p3, t, Dfrag + ((size_t)t.n * m + t.sub) * fr, hist + (size_t)t.n * hstep, PWc, Np);
    double *out = dpsi + ((size_t)t.n * m + t.sub) * hstep;
    #pragma unroll
    for (int r = 0; r < DN_RB; r++) {
        if (t.rb[r] < 0) continue;
        #pragma unroll
        for (int p = 0; p < DN_NG / 2; p++) {
            const int g = lane_group(t, p);
            if (g < 0) continue;
            #pragma unroll
            for (int e = 0; e < 4; e++) {
                double *o = out + (size_t)(t.rb[r] * 16 + t.kk + 4 * e) * PWc + g * 16 + (t.c16 & 7);
                const double s12 = p1[r][p][e] + p2[r][p][e];
                o[0] = p1[r][p][e] - p2[r][p][e];
                o[8] = p3[r][p][e] - s12;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// lambda_n = L_n^{-H} y_n, n = 1..nt-1 (the time point of a tile is n-1); also clears sigma and grad, which the
// gradient kernels accumulate into.  LinvT: planes with L^{-1}(k, row) at [row + Np*k].
// ---------------------------------------------------------------------------
template <int DN_RB, int DN_NG>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_lambda_f(const double *__restrict__ LinvT, const double *__restrict__ yhist,
                                                  double *__restrict__ lam, int Np, int cp, int nt,
                                                  double *__restrict__ zero_a, int n_a, double *__restrict__ zero_b, int n_b)
{
    {
        const int gid = blockIdx.x * blockDim.x + threadIdx.x, gsz = gridDim.x * blockDim.x;
        for (int e = gid; e < n_a; e += gsz) zero_a[e] = 0.0;
        for (int e = gid; e < n_b; e += gsz) zero_b[e] = 0.0;
    }
    DenseTile<DN_RB, DN_NG> t;
    if (!dense_tile(t, Np >> 4, cp >> 3, 1, nt - 1)) return;
    const int n = t.n + 1, PWc = 2 * cp;
    const size_t hstep = (size_t)Np * PWc, pl = (size_t)Np * Np;
    d4 acc[DN_RB][DN_NG];
    ZERO_ACC(acc);
    cgemm_tile_planes_conj(acc, t, LinvT + (size_t)n * 2 * pl, LinvT + (size_t)n * 2 * pl + pl, yhist + (size_t)n * hstep, PWc, Np);
    double *out = lam + (size_t)n * hstep;
    #pragma unroll
    for (int r = 0; r < DN_RB; r++) {
        if (t.rb[r] < 0) continue;
        #pragma unroll
        for (int g = 0; g < DN_NG; g++) {
            if (t.g[g] < 0) continue;
            #pragma unroll
            for (int e = 0; e < 4; e++)
                out[(size_t)(t.rb[r] * 16 + t.kk + 4 * e) * PWc + t.g[g] * 16 + t.c16] = acc[r][g][e];
        }
    }
}

template <int DN_RB, int DN_NG>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_lambda_f3(const double *__restrict__ LinvT, const double *__restrict__ yhist,
                                                  double *__restrict__ lam, int Np, int cp, int nt,
                                                  double *__restrict__ zero_a, int n_a, double *__restrict__ zero_b, int n_b)
{
    {
        const int gid = blockIdx.x * blockDim.x + threadIdx.x, gsz = gridDim.x * blockDim.x;
        for (int e = gid; e < n_a; e += gsz) zero_a[e] = 0.0;
        for (int e = gid; e < n_b; e += gsz) zero_b[e] = 0.0;
    }
    DenseTile<DN_RB, DN_NG> t;
    if (!dense_tile(t, Np >> 4, cp >> 3, 1, nt - 1)) return;
    const int n = t.n + 1, PWc = 2 * cp;
    const size_t hstep = (size_t)Np * PWc, pl = (size_t)Np * Np;
    d4 p1[DN_RB][DN_NG / 2], p2[DN_RB][DN_NG / 2], p3[DN_RB][DN_NG / 2];
    ZERO_ACC3(p1); ZERO_ACC3(p2); ZERO_ACC3(p3);
    cgemm3_tile_planes<true>(p1, p2, p3, t, LinvT + (size_t)n * 2 * pl, LinvT + (size_t)n * 2 * pl + pl, yhist + (size_t)n * hstep, PWc, Np);
    double *out = lam + (size_t)n * hstep;
    #pragma unroll
    for (int r = 0; r < DN_RB; r++) {
        if (t.rb[r] < 0) continue;
        #pragma unroll
        for (int p = 0; p < DN_NG / 2; p++) {
            const int g = lane_group(t, p);
            if (g < 0) continue;
            #pragma unroll
            for (int e = 0; e < 4; e++) {
                double *o = out + (size_t)(t.rb[r] * 16 + t.kk + 4 * e) * PWc + g * 16 + (t.c16 & 7);
                o[0] = p1[r][p][e] + p2[r][p][e];                       // conjugated operand: s = -1
                o[8] = p3[r][p][e] - p1[r][p][e] + p2[r][p][e];
            }
        }
    }
}

// P[n] = Linv[n+1] R[n] on the three-product tiles (k_propagator in qgd_k_inverse.hip is the four-product kernel): panel and
// column-major planes as the chains want them
template <int DN_RB, int DN_NG>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_propagator3(const double *__restrict__ LinvA, const double *__restrict__ R,
                                                  double *__restrict__ Pr, double *__restrict__ Pc, int Np, int nt)
{
    DenseTile<DN_RB, DN_NG> t;
    if (!dense_tile(t, Np >> 4, Np >> 3, 1, nt - 1)) return;
    const int n = t.n, PW = 2 * Np;
    const size_t panel = (size_t)Np * PW, pl = (size_t)Np * Np;
    d4 p1[DN_RB][DN_NG / 2], p2[DN_RB][DN_NG / 2], p3[DN_RB][DN_NG / 2];
    ZERO_ACC3(p1); ZERO_ACC3(p2); ZERO_ACC3(p3);
    cgemm3_tile_planes<false>(p1, p2, p3, t, LinvA + (size_t)(n + 1) * 2 * pl, LinvA + (size_t)(n + 1) * 2 * pl + pl, R + (size_t)n * panel, PW, Np);
    double *Prn = Pr + (size_t)n * panel, *Pcn = Pc + (size_t)n * 2 * pl;
    #pragma unroll
    for (int r = 0; r < DN_RB; r++) {
        if (t.rb[r] < 0) continue;
        #pragma unroll
        for (int p = 0; p < DN_NG / 2; p++) {
            const int g = lane_group(t, p);
            if (g < 0) continue;
            const int ccol = g * 8 + (t.c16 & 7);
            #pragma unroll
            for (int e = 0; e < 4; e++) {
                const int row = t.rb[r] * 16 + t.kk + 4 * e;
                const double s12 = p1[r][p][e] + p2[r][p][e];
                const double vre = p1[r][p][e] - p2[r][p][e], vim = p3[r][p][e] - s12;
                double *o = Prn + (size_t)row * PW + g * 16 + (t.c16 & 7);
                o[0] = vre; o[8] = vim;
                Pcn[(size_t)row + (size_t)Np * ccol] = vre;
                Pcn[pl + (size_t)row + (size_t)Np * ccol] = vim;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Inverse of the N x N step matrices (N > 64) as BLOCK Gauss-Jordan over 64-column blocks, the block operations spread over
// the whole chip as batched three-product GEMM tiles (k_inverse_blocked2 keeps one matrix on one CU: 201 matrices at config
// 5 are 201 CUs running a pivot-latency-bound kernel for 1.8 ms, on any number of GPUs).  Step k = 0, 1, ... on W (W = L first):
//     D    = inv(W_kk)                                   k_inverse_diag: partial pivoting INSIDE the 64 x 64 block
//     W'_k. = D [W_k. with block k replaced by I]         k_binv_row    (64 x 64 x N)
//     W'_i. = [W_i. with block k zeroed] - W_ik W'_k.      k_binv_rest   ((N - 64) x 64 x N)
// and W is the inverse after the last block.  Out of place (two panel-layout buffers alternate); the left operands W_ik
// are read from column-major planes that k_binv_planes transposes out of the panel layout for the 64 columns a step
// needs (LinvA serves as that buffer until the final transpose fills it).  No pivoting ACROSS blocks: fine when the
// block multipliers W_ik D stay small -- L = I + i dt/2 H - ... has leading blocks as well conditioned as itself for
// the time steps these schemes are accurate at -- and checked: an entry of a multiplier block beyond `thresh` (or a zero
// pivot inside a diagonal block) marks the matrix, and marked matrices are redone by k_inverse_blocked2 with full partial
// pivoting from the untouched L in the same stream (a launch that ends at once for the others).
// ---------------------------------------------------------------------------
#define BINV_B 64
template <int DN_RB, int DN_NG>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_binv_row(const double *__restrict__ DkC, const double *__restrict__ Win,
                                                  size_t in_stride, double *__restrict__ Wout, int Np, int nmat, int kb0, int bs)
{
    DenseTile<DN_RB, DN_NG> t;
    if (!dense_tile(t, bs >> 4, Np >> 3, 1, nmat)) return;
    const int n = t.n + 1, PW = 2 * Np;
    const size_t panel = (size_t)Np * PW;
    const double *A = DkC + (size_t)n * 2 * BINV_B * BINV_B;
    d4 p1[DN_RB][DN_NG / 2], p2[DN_RB][DN_NG / 2], p3[DN_RB][DN_NG / 2];
    ZERO_ACC3(p1); ZERO_ACC3(p2); ZERO_ACC3(p3);
    cgemm3_tile_planes<false>(p1, p2, p3, t, A, A + BINV_B * BINV_B, Win + (size_t)n * in_stride + (size_t)kb0 * PW, PW, BINV_B, bs);
    double *out = Wout + (size_t)n * panel;
    #pragma unroll
    for (int r = 0; r < DN_RB; r++) {
        if (t.rb[r] < 0) continue;
        #pragma unroll
        for (int p = 0; p < DN_NG / 2; p++) {
            const int g = lane_group(t, p);
            if (g < 0) continue;
            const int ccol = g * 8 + (t.c16 & 7);
            const bool in_blk = ccol >= kb0 && ccol < kb0 + bs;
            #pragma unroll
            for (int e = 0; e < 4; e++) {
                const int lrow = t.rb[r] * 16 + t.kk + 4 * e;
                const double s12 = p1[r][p][e] + p2[r][p][e];
                double vre = p1[r][p][e] - p2[r][p][e], vim = p3[r][p][e] - s12;
                if (in_blk) { vre = A[lrow + BINV_B * (ccol - kb0)]; vim = A[BINV_B * BINV_B + lrow + BINV_B * (ccol - kb0)]; }
                double *o = out + (size_t)(kb0 + lrow) * PW + g * 16 + (t.c16 & 7);
                o[0] = vre; o[8] = vim;
            }
        }
    }
}

template <int DN_RB, int DN_NG>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_binv_rest(const double *__restrict__ Cin, const double *__restrict__ Win,
                                                  size_t in_stride, double *__restrict__ Wout, int Np, int nmat, int kb0, int bs,
                                                  int *__restrict__ flags, double thresh)
{
    DenseTile<DN_RB, DN_NG> t;
    if (!dense_tile(t, (Np - bs) >> 4, Np >> 3, 1, nmat)) return;
    #pragma unroll
    for (int r = 0; r < DN_RB; r++) if (t.rb[r] >= (kb0 >> 4)) t.rb[r] += bs >> 4;       // the row blocks outside block k
    const int n = t.n + 1, PW = 2 * Np;
    const size_t panel = (size_t)Np * PW, pl = (size_t)Np * Np;
    const double *Are = Cin + (size_t)n * 2 * pl + (size_t)Np * kb0;
    d4 p1[DN_RB][DN_NG / 2], p2[DN_RB][DN_NG / 2], p3[DN_RB][DN_NG / 2];
    ZERO_ACC3(p1); ZERO_ACC3(p2); ZERO_ACC3(p3);
    double *out = Wout + (size_t)n * panel;
    cgemm3_tile_planes<false>(p1, p2, p3, t, Are, Are + pl, out + (size_t)kb0 * PW, PW, Np, bs);
    const double *in = Win + (size_t)n * in_stride;
    bool bad = false;
    #pragma unroll
    for (int r = 0; r < DN_RB; r++) {
        if (t.rb[r] < 0) continue;
        #pragma unroll
        for (int p = 0; p < DN_NG / 2; p++) {
            const int g = lane_group(t, p);
            if (g < 0) continue;
            const int ccol = g * 8 + (t.c16 & 7);
            const bool in_blk = ccol >= kb0 && ccol < kb0 + bs;
            #pragma unroll
            for (int e = 0; e < 4; e++) {
                const size_t o = (size_t)(t.rb[r] * 16 + t.kk + 4 * e) * PW + g * 16 + (t.c16 & 7);
                const double s12 = p1[r][p][e] + p2[r][p][e];
                const double bre = in_blk ? 0.0 : in[o], bim = in_blk ? 0.0 : in[o + 8];
                const double vre = bre - (p1[r][p][e] - p2[r][p][e]), vim = bim - (p3[r][p][e] - s12);
                if (in_blk) bad = bad || !(fabs(vre) <= thresh && fabs(vim) <= thresh);      // multiplier block -W_ik D (NaN counts)
                out[o] = vre; out[o + 8] = vim;
            }
        }
    }
    if (bad) flags[n] = 1;
}

// panel layout -> column-major planes C[row + Np col] (+ Np^2: imaginary parts) for the columns [col0, col0 + nc) of every
// matrix, and (T != null) row-major planes T[row Np + col].  32 x 32 tiles through LDS, both sides in runs of 8-32 doubles.
// grid ceil(Np/32) * ceil(nc/32) * nmat (flat: a long grid may have more than 65535 matrices)
__global__ __launch_bounds__(256) void k_binv_planes(const double *__restrict__ Win, size_t in_stride, double *__restrict__ C, double *__restrict__ T,
                                                     int Np, int col0, int nc)
{
    __shared__ double tre[32][33], tim[32][33];
    const int PW = 2 * Np;
    const size_t pl = (size_t)Np * Np;
    const int ctiles = (nc + 31) / 32, tiles = ((Np + 31) / 32) * ctiles;
    const int n = (int)(blockIdx.x / tiles) + 1, tl = (int)(blockIdx.x % tiles);
    const int r0 = (tl / ctiles) * 32, c0 = col0 + (tl % ctiles) * 32;
    const double *in = Win + (size_t)n * in_stride;
    #pragma unroll
    for (int k = 0; k < 4; k++) {
        const int idx = threadIdx.x + k * 256, rl = idx >> 5, cl = idx & 31, row = r0 + rl, col = c0 + cl;
        if (row < Np && col < col0 + nc) {
            const size_t o = (size_t)row * PW + (col >> 3) * 16 + (col & 7);
            tre[rl][cl] = in[o]; tim[rl][cl] = in[o + 8];
        }
    }
    __syncthreads();
    double *Cn = C + (size_t)n * 2 * pl;
    #pragma unroll
    for (int k = 0; k < 4; k++) {
        const int idx = threadIdx.x + k * 256;
        {
            const int cl = idx >> 5, rl = idx & 31, row = r0 + rl, col = c0 + cl;
            if (row < Np && col < col0 + nc) { Cn[(size_t)row + (size_t)Np * col] = tre[rl][cl]; Cn[pl + (size_t)row + (size_t)Np * col] = tim[rl][cl]; }
        }
        if (T) {
            const int rl = idx >> 5, cl = idx & 31, row = r0 + rl, col = c0 + cl;
            if (row < Np && col < col0 + nc) { T[(size_t)n * 2 * pl + (size_t)row * Np + col] = tre[rl][cl]; T[(size_t)n * 2 * pl + pl + (size_t)row * Np + col] = tim[rl][cl]; }
        }
    }
}

// ---------------------------------------------------------------------------
// gradient: seeds g_j = c_j dt^j lambda_{n+1} [n <= nt-2] - c_j (-dt)^j lambda_n [n >= 1], j = 1..m
// Gp: [nt][m] panels [Np][2cp].  grid (ceil(hstep/256), nt)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_ginit(const double *__restrict__ lam, const double *__restrict__ cw,
                                               double *__restrict__ Gp, size_t hstep, int m, int nt)
{
    const int n = blockIdx.y;
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= hstep) return;
    const double ln = (n >= 1) ? lam[(size_t)n * hstep + e] : 0.0;
    const double lx = (n <= nt - 2) ? lam[(size_t)(n + 1) * hstep + e] : 0.0;
    for (int j = 1; j <= m; j++) Gp[((size_t)n * m + (j - 1)) * hstep + e] = cw[2 * j] * lx - cw[2 * j + 1] * ln;
}

// reverse sweep, level j: g_i -= (1/j) A_{j-1-i} g_j for i = 1..j-1 (sub-index = i-1); A^H = -A
template <int DN_RB, int DN_NG>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_gsweep_f(const d2 *__restrict__ Afrag, double *__restrict__ Gp, int Np, int cp,
                                                  int m, int nt, int j)
{
    DenseTile<DN_RB, DN_NG> t;
    if (!dense_tile(t, Np >> 4, cp >> 3, j - 1, nt, true)) return;
    const int i = t.sub + 1;
    const int PWc = 2 * cp;
    const size_t hstep = (size_t)Np * PWc, fr = (size_t)Np * Np;
    d4 acc[DN_RB][DN_NG];
    ZERO_ACC(acc);
    cgemm_tile(acc, t, Afrag + ((size_t)t.n * m + (j - 1 - i)) * fr, Gp + ((size_t)t.n * m + (j - 1)) * hstep, PWc, Np);
    double *out = Gp + ((size_t)t.n * m + (i - 1)) * hstep;
    const double sc = -1.0 / (double)j;
    #pragma unroll
    for (int r = 0; r < DN_RB; r++) {
        if (t.rb[r] < 0) continue;
        #pragma unroll
        for (int g = 0; g < DN_NG; g++) {
            if (t.g[g] < 0) continue;
            #pragma unroll
            for (int e = 0; e < 4; e++)
                out[(size_t)(t.rb[r] * 16 + t.kk + 4 * e) * PWc + t.g[g] * 16 + t.c16] += sc * acc[r][g][e];
        }
    }
}

template <int DN_RB, int DN_NG>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_gsweep_f3(const d2 *__restrict__ Afrag, double *__restrict__ Gp, int Np, int cp,
                                                  int m, int nt, int j, d2 *__restrict__ Gfrag, const double *__restrict__ Tseed,
                                                  const double *__restrict__ cw)
{
    DenseTile<DN_RB, DN_NG> t;
    if (!dense_tile(t, Np >> 4, cp >> 3, j - 1, nt, true)) return;
    const int i = t.sub + 1;
    const int PWc = 2 * cp;
    const size_t hstep = (size_t)Np * PWc, fr = (size_t)Np * Np;
    d4 p1[DN_RB][DN_NG / 2], p2[DN_RB][DN_NG / 2], p3[DN_RB][DN_NG / 2];
    ZERO_ACC3(p1); ZERO_ACC3(p2); ZERO_ACC3(p3);
    cgemm3_tile(p1, p2, p3, t, Afrag + ((size_t)t.n * m + (j - 1 - i)) * fr, Gp + ((size_t)t.n * m + (j - 1)) * hstep, PWc, Np);
    double *out = Gp + ((size_t)t.n * m + (i - 1)) * hstep;
    const double sc = -1.0 / (double)j;
    #pragma unroll
    for (int r = 0; r < DN_RB; r++) {
        if (t.rb[r] < 0) continue;
        #pragma unroll
        for (int p = 0; p < DN_NG / 2; p++) {
            const int g = lane_group(t, p);
            if (g < 0) continue;
            #pragma unroll
            for (int e = 0; e < 4; e++) {
                const int row = t.rb[r] * 16 + t.kk + 4 * e;
                double *o = out + (size_t)row * PWc + g * 16 + (t.c16 & 7);
                const double s12 = p1[r][p][e] + p2[r][p][e];
                double bre, bim;
                if (Tseed) {     // first level (j = m) of the third gradient form: the seed Y_i = c_i dt^i Lambda+ - c_i (-dt)^i Lambda- is
                                 // formed here from the two outer products instead of being written and read back (k_yinit)
                    const size_t to = (size_t)t.n * 2 * hstep + (size_t)row * PWc + g * 16 + (t.c16 & 7);
                    const double wx = (t.n <= nt - 2) ? cw[2 * i] : 0.0, wn = (t.n >= 1) ? cw[2 * i + 1] : 0.0;
                    bre = wx * Tseed[to] - wn * Tseed[to + hstep];
                    bim = wx * Tseed[to + 8] - wn * Tseed[to + hstep + 8];
                } else { bre = o[0]; bim = o[8]; }
                const double vre = bre + sc * (p1[r][p][e] - p2[r][p][e]), vim = bim + sc * (p3[r][p][e] - s12);
                o[0] = vre; o[8] = vim;
                // level j is the last one that touches g_{j-1}: its final value also in fragment order (square panels only:
                // the matrices Y_j of the third gradient form, left operands of k_ginner_d)
                if (Gfrag && i == j - 1) Gfrag[((size_t)t.n * m + (i - 1)) * fr + frag_index(Np, row, g * 8 + (t.c16 & 7))] = (d2){vre, vim};
            }
        }
    }
}

// Fixed-order reduction of the gradient scalars (round 3: the N > 64 kernels are bitwise reproducible too).  Every wave adds
// into its OWN LDS slots sigw[wave][.] (wave_sig); after the barrier thread e adds the four slots in order and STORES the
// sum into the workgroup's own plane of sigma -- plane = the tile's slot inside its time point -- and k_contract adds the
// planes in order.  No atomics: every (plane, n, o, d) entry has exactly one writer.
__device__ __forceinline__ double *wave_sig(double *sig, int nvals) { return sig + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) * nvals; }
__device__ __forceinline__ double sum_wave_sig(const double *sig, int nvals, int e)
{
    return ((sig[e] + sig[nvals + e]) + sig[2 * nvals + e]) + sig[3 * nvals + e];
}

// sigma[n][o][d][2] += (1/j) < (dA_d/d{p,q}_o) psi_i, g_j >, j = i+1+d, for the source level i (sub-index)
template <int DN_RB, int DN_NG>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_ginner_f(const d2 *__restrict__ OpFrag, const double *__restrict__ hist,
                                                  const double *__restrict__ dpsi, const double *__restrict__ Gp,
                                                  double *__restrict__ sigma, int Np, int cp, int n_ops, int m, int nt)
{
    extern __shared__ double sig[];          // [4 waves][n_ops][m][2]
    for (int e = threadIdx.x; e < 4 * n_ops * m * 2; e += blockDim.x) sig[e] = 0.0;
    __syncthreads();
    DenseTile<DN_RB, DN_NG> t;
    const bool active = dense_tile(t, Np >> 4, cp >> 3, m, nt, true);
    if (active) {
        double *sw = wave_sig(sig, n_ops * m * 2);
        const int i = t.sub;
        const int PWc = 2 * cp;
        const size_t hstep = (size_t)Np * PWc, fr = (size_t)Np * Np;
        const double *src = (i == 0) ? hist + (size_t)t.n * hstep : dpsi + ((size_t)t.n * m + (i - 1)) * hstep;
        for (int o = 0; o < n_ops; o++) {
            d4 U[DN_RB][DN_NG], V[DN_RB][DN_NG];
            ZERO_ACC(U);
            ZERO_ACC(V);
            rgemm2_tile(U, V, t, OpFrag + (size_t)o * fr, src, PWc, Np);
            _Pragma("unroll 1")
            for (int j = i + 1; j <= m; j++) {
                const double *gj = Gp + ((size_t)t.n * m + (j - 1)) * hstep;
                double sp = 0.0, sq = 0.0;
                #pragma unroll
                for (int r = 0; r < DN_RB; r++) {
                    if (t.rb[r] < 0) continue;
                    #pragma unroll
                    for (int g = 0; g < DN_NG; g++) {
                        if (t.g[g] < 0) continue;
                        #pragma unroll
                        for (int e = 0; e < 4; e++) {
                            const double g1 = gj[(size_t)(t.rb[r] * 16 + t.kk + 4 * e) * PWc + t.g[g] * 16 + t.c16];
                            const double g2 = swap8_signed(g1, t.sign_hi);
                            sq += V[r][g][e] * g1;       // Re <K psi, g>
                            sp += U[r][g][e] * g2;       // Re <-i S psi, g>
                        }
                    }
                }
                for (int off = 32; off > 0; off >>= 1) { sp += __shfl_down(sp, off); sq += __shfl_down(sq, off); }
                if (t.lane == 0) {
                    const int d = j - 1 - i;
                    atomicAdd(&sw[(o * m + d) * 2], sp / (double)j);       // this wave's own slots: one writer, and LDS executes a wave's
                    atomicAdd(&sw[(o * m + d) * 2 + 1], sq / (double)j);   // operations in order (the add without a return value does not stall)
                }
            }
        }
    }
    __syncthreads();
    if (!active && t.n >= nt) return;
    // plane = the tile's slot (source level i included): entries d >= m - i of it are zero (sum_wave_sig of untouched slots)
    const size_t plane = (size_t)nt * n_ops * m * 2;
    for (int e = threadIdx.x; e < n_ops * m * 2; e += blockDim.x)
        sigma[(size_t)t.slot * plane + (size_t)t.n * n_ops * m * 2 + e] = sum_wave_sig(sig, n_ops * m * 2, e);
}

// The same scalars from OUTER products over the columns instead of operator applications:
//   sigma^P[o][d] = sum_i (1/j) Re<-i S_o psi_i, g_j>,  sigma^Q[o][d] = sum_i (1/j) Re<K_o psi_i, g_j>,  j = i + 1 + d
// with S_o, K_o real:  Re<K psi, g> = <K, Re M>_F and Re<-i S psi, g> = -<S, Im M>_F for the N x N matrix
//   M(i, j)[r1, r2] = sum_col g_j[r1, col] conj(psi_i[r2, col]),
// so per time point  M_d = sum_i (1/j) M(i, i+1+d), d = 0..m-1, costs m(m+1)/2 products N x N x c -- independent of the
// number of control operators -- against n_ops * m for k_ginner_f, and sigma[.][d] is 2 n_ops Frobenius products with
// M_d in the epilogue.  Used when (m+1)/2 < n_ops (config 5: 21 instead of 24 GEMM units, the largest kernel of the
// evaluation).  Tile: a wave owns DN_RB x DN_NG blocks of 16 x 16 of M_d (rows r1 from g, columns r2 from psi), the
// contraction runs over the columns of the state panels: both operands are read as [row][8 re | 8 im] groups, a lane
// takes two adjacent columns in one 16-byte load and feeds them to two MFMAs (the contraction order is free as long
// as both operands use the same one).  Three accumulators per block: Re M, sum g_im psi_re, sum g_re psi_im.  The
// weights 1/j differ from pair to pair: the running sum is rescaled by j/(j+1)... (w_prev/w_next) between pairs.
// (aR, aA, aB) += { Re, sum A_im B_re, sum A_re B_im } of  A B^H  for the wave's blocks: A, B row-major [row][groups of 8 re | 8 im]
// with row stride ld doubles, contraction over ngc groups of 8 columns.  A lane takes two adjacent columns per 16-byte
// load and feeds them to two MFMAs (the contraction order is free as long as both operands use the same one); the
// operands of group q+1 are in flight while the 32 MFMAs of group q issue.  Im(A B^H) = aA - aB.
template <int DN_RB, int DN_NG>
__device__ __forceinline__ void outer_tile(d4 (&aR)[DN_RB][DN_NG], d4 (&aA)[DN_RB][DN_NG], d4 (&aB)[DN_RB][DN_NG],
                                           const DenseTile<DN_RB, DN_NG> &t, const double *__restrict__ A, const double *__restrict__ B,
                                           size_t ld, int ngc)
{
    size_t ao[DN_RB], bo[DN_NG];
    #pragma unroll
    for (int r = 0; r < DN_RB; r++) ao[r] = (size_t)((t.rb[r] >= 0 ? t.rb[r] : t.rb[0]) * 16 + t.c16) * ld + 2 * t.kk;
    #pragma unroll
    for (int g = 0; g < DN_NG; g++) bo[g] = (size_t)((t.g[g] >= 0 ? t.g[g] : t.g[0]) * 16 + t.c16) * ld + 2 * t.kk;
    d2 are[DN_RB], aim[DN_RB], bre[DN_NG], bim[DN_NG], aren[DN_RB], aimn[DN_RB], bren[DN_NG], bimn[DN_NG];
    #pragma unroll
    for (int r = 0; r < DN_RB; r++) { are[r] = *reinterpret_cast<const d2 *>(A + ao[r]); aim[r] = *reinterpret_cast<const d2 *>(A + ao[r] + 8); }
    #pragma unroll
    for (int g = 0; g < DN_NG; g++) { bre[g] = *reinterpret_cast<const d2 *>(B + bo[g]); bim[g] = *reinterpret_cast<const d2 *>(B + bo[g] + 8); }
    for (int q = 0; q < ngc; q++) {
        const size_t qn = (size_t)((q + 1 < ngc) ? q + 1 : q) * 16;
        #pragma unroll
        for (int r = 0; r < DN_RB; r++) { aren[r] = *reinterpret_cast<const d2 *>(A + ao[r] + qn); aimn[r] = *reinterpret_cast<const d2 *>(A + ao[r] + qn + 8); }
        #pragma unroll
        for (int g = 0; g < DN_NG; g++) { bren[g] = *reinterpret_cast<const d2 *>(B + bo[g] + qn); bimn[g] = *reinterpret_cast<const d2 *>(B + bo[g] + qn + 8); }
        #pragma unroll
        for (int s_ = 0; s_ < 2; s_++) {
            #pragma unroll
            for (int r = 0; r < DN_RB; r++)
                #pragma unroll
                for (int g = 0; g < DN_NG; g++) aR[r][g] = MFMA(are[r][s_], bre[g][s_], aR[r][g]);
            #pragma unroll
            for (int r = 0; r < DN_RB; r++)
                #pragma unroll
                for (int g = 0; g < DN_NG; g++) aA[r][g] = MFMA(aim[r][s_], bre[g][s_], aA[r][g]);
            #pragma unroll
            for (int r = 0; r < DN_RB; r++)
                #pragma unroll
                for (int g = 0; g < DN_NG; g++) aR[r][g] = MFMA(aim[r][s_], bim[g][s_], aR[r][g]);
            #pragma unroll
            for (int r = 0; r < DN_RB; r++)
                #pragma unroll
                for (int g = 0; g < DN_NG; g++) aB[r][g] = MFMA(are[r][s_], bim[g][s_], aB[r][g]);
        }
        #pragma unroll
        for (int r = 0; r < DN_RB; r++) { are[r] = aren[r]; aim[r] = aimn[r]; }
        #pragma unroll
        for (int g = 0; g < DN_NG; g++) { bre[g] = bren[g]; bim[g] = bimn[g]; }
    }
}

// The same product with three real products per complex multiply (see cgemm3_tile):  A B^H = (Ar Br + Ai Bi) + i (Ai Br - Ar Bi),
//     P1 = Ar Br,  P2 = Ai Bi,  P3 = (Ar + Ai)(Br - Bi);   Re = P1 + P2,  Im = P3 - P1 + P2.
// Both operands already have their real and imaginary parts in different registers, so nothing but the two operand
// sums is added: 24 MFMAs instead of 32 per group of 8 contraction columns of a <2, 2> tile.  Buffer-addressed loads and
// two register sets as in cgemm3_tile.
template <int DN_RB, int DN_NG>
__device__ __forceinline__ void outer_tile3(d4 (&p1)[DN_RB][DN_NG], d4 (&p2)[DN_RB][DN_NG], d4 (&p3)[DN_RB][DN_NG],
                                            const DenseTile<DN_RB, DN_NG> &t, const double *__restrict__ A, const double *__restrict__ B,
                                            size_t ld, int ngc)
{
    const __amdgpu_buffer_rsrc_t ra = buffer_of(A), rb_ = buffer_of(B);
    int av[DN_RB], bv[DN_NG];
    #pragma unroll
    for (int r = 0; r < DN_RB; r++) av[r] = (((t.rb[r] >= 0 ? t.rb[r] : t.rb[0]) * 16 + t.c16) * (int)ld + 2 * t.kk) * 8;
    #pragma unroll
    for (int g = 0; g < DN_NG; g++) bv[g] = (((t.g[g] >= 0 ? t.g[g] : t.g[0]) * 16 + t.c16) * (int)ld + 2 * t.kk) * 8;
    // (the loop body must stay free of branches: with a conditional step inside, the compiler's wait-count pass falls back to
    //  s_waitcnt vmcnt(0) in front of every group and the prefetch is gone -- seen in the ISA, 0.61 MFMA busy)
    d2 ar0[DN_RB], ai0[DN_RB], br0[DN_NG], bi0[DN_NG], ar1[DN_RB], ai1[DN_RB], br1[DN_NG], bi1[DN_NG];
#define O3_LOAD(ar, ai, br, bi, q) do {                                                                                          \
        _Pragma("unroll") for (int r = 0; r < DN_RB; r++) { ar[r] = buffer_load_d2(ra, av[r], (q) * 128); ai[r] = buffer_load_d2(ra, av[r] + 64, (q) * 128); }  \
        _Pragma("unroll") for (int g = 0; g < DN_NG; g++) { br[g] = buffer_load_d2(rb_, bv[g], (q) * 128); bi[g] = buffer_load_d2(rb_, bv[g] + 64, (q) * 128); } } while (0)
#define O3_MFMA(ar, ai, br, bi) do {                                                                                             \
        d2 as[DN_RB], bd[DN_NG];                                                                                                 \
        _Pragma("unroll") for (int r = 0; r < DN_RB; r++) as[r] = ar[r] + ai[r];                                                   \
        _Pragma("unroll") for (int g = 0; g < DN_NG; g++) bd[g] = br[g] - bi[g];                                                   \
        _Pragma("unroll") for (int s_ = 0; s_ < 2; s_++) {                                                                        \
            _Pragma("unroll") for (int r = 0; r < DN_RB; r++)                                                                     \
                _Pragma("unroll") for (int g = 0; g < DN_NG; g++) p1[r][g] = MFMA(ar[r][s_], br[g][s_], p1[r][g]);                 \
            _Pragma("unroll") for (int r = 0; r < DN_RB; r++)                                                                     \
                _Pragma("unroll") for (int g = 0; g < DN_NG; g++) p2[r][g] = MFMA(ai[r][s_], bi[g][s_], p2[r][g]);                 \
            _Pragma("unroll") for (int r = 0; r < DN_RB; r++)                                                                     \
                _Pragma("unroll") for (int g = 0; g < DN_NG; g++) p3[r][g] = MFMA(as[r][s_], bd[g][s_], p3[r][g]);                 \
        } } while (0)
    O3_LOAD(ar0, ai0, br0, bi0, 0);
    const int npair = ngc >> 1;
    for (int q = 0; q < 2 * npair; q += 2) {
        O3_LOAD(ar1, ai1, br1, bi1, q + 1);
        O3_MFMA(ar0, ai0, br0, bi0);
        const int q2 = (q + 2 < ngc) ? q + 2 : q;
        O3_LOAD(ar0, ai0, br0, bi0, q2);
        O3_MFMA(ar1, ai1, br1, bi1);
    }
    if (ngc & 1) O3_MFMA(ar0, ai0, br0, bi0);        // (an odd group count: the last group was loaded by the final q2)
#undef O3_LOAD
#undef O3_MFMA
}

// ... and with BOTH operands in fragment order (frag[(rb * K/4 + k4) * 64 + lane] = {Re, Im} of element (16 rb + lane % 16,
// 4 k4 + lane / 16), the layout k_level_f writes the D_i in): every load is one contiguous KB per wave.  With the row-major
// panels of outer_tile3 the 16 lanes of an MFMA row are 16 rows 4 KB apart -- 64 cache-line lookups per load instruction,
// and k_ginner_d sat on the L1's tag rate (MFMA pipe busy 0.64 after the 3M change).
template <int DN_RB, int DN_NG>
__device__ __forceinline__ void outer_frag3(d4 (&p1)[DN_RB][DN_NG], d4 (&p2)[DN_RB][DN_NG], d4 (&p3)[DN_RB][DN_NG],
                                            const DenseTile<DN_RB, DN_NG> &t, const d2 *__restrict__ A, const d2 *__restrict__ B, int K)
{
    const __amdgpu_buffer_rsrc_t ra = buffer_of(reinterpret_cast<const double *>(A)), rb_ = buffer_of(reinterpret_cast<const double *>(B));
    int av[DN_RB], bv[DN_NG];
    #pragma unroll
    for (int r = 0; r < DN_RB; r++) av[r] = ((t.rb[r] >= 0 ? t.rb[r] : t.rb[0]) * (K >> 2) * 64 + t.lane) * 16;
    #pragma unroll
    for (int g = 0; g < DN_NG; g++) bv[g] = ((t.g[g] >= 0 ? t.g[g] : t.g[0]) * (K >> 2) * 64 + t.lane) * 16;
    const int nk4 = K >> 2;              // (K is a multiple of 16: nk4 is even)
    d2 a0[DN_RB], b0[DN_NG], a1[DN_RB], b1[DN_NG];
#define F3_LOAD(a, b, k) do {                                                                            \
        _Pragma("unroll") for (int r = 0; r < DN_RB; r++) a[r] = buffer_load_d2(ra, av[r], (k) * 1024);    \
        _Pragma("unroll") for (int g = 0; g < DN_NG; g++) b[g] = buffer_load_d2(rb_, bv[g], (k) * 1024); } while (0)
#define F3_MFMA(a, b) do {                                                                               \
        double as[DN_RB], bd[DN_NG];                                                                     \
        _Pragma("unroll") for (int r = 0; r < DN_RB; r++) as[r] = a[r].x + a[r].y;                         \
        _Pragma("unroll") for (int g = 0; g < DN_NG; g++) bd[g] = b[g].x - b[g].y;                         \
        _Pragma("unroll") for (int r = 0; r < DN_RB; r++)                                                 \
            _Pragma("unroll") for (int g = 0; g < DN_NG; g++) {                                           \
                p1[r][g] = MFMA(a[r].x, b[g].x, p1[r][g]);                                                \
                p2[r][g] = MFMA(a[r].y, b[g].y, p2[r][g]);                                                \
                p3[r][g] = MFMA(as[r], bd[g], p3[r][g]);                                                  \
            } } while (0)
    F3_LOAD(a0, b0, 0);
    for (int k4 = 0; k4 < nk4; k4 += 2) {
        F3_LOAD(a1, b1, k4 + 1);
        F3_MFMA(a0, b0);
        const int kn = (k4 + 2 < nk4) ? k4 + 2 : k4;
        F3_LOAD(a0, b0, kn);
        F3_MFMA(a1, b1);
    }
#undef F3_LOAD
#undef F3_MFMA
}

// accumulators of either tile routine -> (Re, Im) of A B^H in (x, y);  M3: (P1, P2, P3), else (Re, sum A_im B_re, sum A_re B_im)
template <bool M3, int DN_RB, int DN_NG>
__device__ __forceinline__ void outer_finish(d4 (&x)[DN_RB][DN_NG], d4 (&y)[DN_RB][DN_NG], const d4 (&z)[DN_RB][DN_NG])
{
    #pragma unroll
    for (int r = 0; r < DN_RB; r++)
        #pragma unroll
        for (int g = 0; g < DN_NG; g++) {
            if (M3) { const d4 re = x[r][g] + y[r][g], im = z[r][g] - x[r][g] + y[r][g]; x[r][g] = re; y[r][g] = im; }
            else y[r][g] = y[r][g] - z[r][g];
        }
}
template <bool M3, int DN_RB, int DN_NG>
__device__ __forceinline__ void outer_any(d4 (&x)[DN_RB][DN_NG], d4 (&y)[DN_RB][DN_NG], d4 (&z)[DN_RB][DN_NG],
                                          const DenseTile<DN_RB, DN_NG> &t, const double *__restrict__ A, const double *__restrict__ B,
                                          size_t ld, int ngc)
{
    if (M3) outer_tile3(x, y, z, t, A, B, ld, ngc); else outer_tile(x, y, z, t, A, B, ld, ngc);
}

// sigma[n][o][d][2] += { -<S_o, Im M>, <K_o, Re M> } * w for the wave's blocks of M (Re = aR, Im = aI).  Element e of block
// (r, g): r1 = rb*16 + kk + 4e, r2 = g*16 + c16.  S_o is symmetric and K_o antisymmetric (SchrodingerProb.jl:73-95): they are
// read TRANSPOSED, [r2 + Np r1] of the column-major planes, so that the 16 lanes of a row read 16 consecutive doubles.
template <int DN_RB, int DN_NG>
__device__ __forceinline__ void frobenius_sigma(const d4 (&aR)[DN_RB][DN_NG], const d4 (&aI)[DN_RB][DN_NG],
                                                const DenseTile<DN_RB, DN_NG> &t, const double *__restrict__ ops, int Np, int n_ops,
                                                double w, double *sig)
{
    const size_t pl = (size_t)Np * Np;
    for (int o = 0; o < n_ops; o++) {
        const double *Ko = ops + (size_t)(2 + 2 * o) * pl, *So = ops + (size_t)(3 + 2 * o) * pl;
        double sp = 0.0, sq = 0.0;
        #pragma unroll
        for (int r = 0; r < DN_RB; r++) {
            if (t.rb[r] < 0) continue;
            #pragma unroll
            for (int g = 0; g < DN_NG; g++) {
                if (t.g[g] < 0) continue;
                #pragma unroll
                for (int e = 0; e < 4; e++) {
                    const size_t at = (size_t)(t.g[g] * 16 + t.c16) + (size_t)Np * (t.rb[r] * 16 + t.kk + 4 * e);
                    sq -= Ko[at] * aR[r][g][e];                       // K[r1,r2] = -K[r2,r1]
                    sp -= So[at] * aI[r][g][e];                       // -<S, Im M>
                }
            }
        }
        for (int off = 32; off > 0; off >>= 1) { sp += __shfl_down(sp, off); sq += __shfl_down(sq, off); }
        if (t.lane == 0) { atomicAdd(&sig[o * 2], sp * w); atomicAdd(&sig[o * 2 + 1], sq * w); }      // (sig: this wave's own slots -- one writer, in order)
    }
}

// The same scalars from OUTER products over the columns instead of operator applications:
//   sigma^P[o][d] = sum_i (1/j) Re<-i S_o psi_i, g_j>,  sigma^Q[o][d] = sum_i (1/j) Re<K_o psi_i, g_j>,  j = i + 1 + d
// with S_o, K_o real:  Re<K psi, g> = <K, Re M>_F and Re<-i S psi, g> = -<S, Im M>_F for the N x N matrix
//   M(i, j)[r1, r2] = sum_col g_j[r1, col] conj(psi_i[r2, col]),
// so per time point  M_d = sum_i (1/j) M(i, i+1+d), d = 0..m-1, costs m(m+1)/2 products N x N x c -- independent of the
// number of control operators -- against n_ops * m for k_ginner_f, and sigma[.][d] is 2 n_ops Frobenius products with
// M_d in the epilogue.  Used when (m+1)/2 < n_ops.  Tile: a wave owns DN_RB x DN_NG blocks of 16 x 16 of M_d (rows r1
// from g, columns r2 from psi).  The weights 1/j differ from pair to pair: the running sum is kept in units of the
// current weight (times w_prev / w_next between pairs).
template <int DN_RB, int DN_NG, bool M3>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_ginner_m(const double *__restrict__ ops, const double *__restrict__ hist,
                                                  const double *__restrict__ dpsi, const double *__restrict__ Gp,
                                                  double *__restrict__ sigma, int Np, int cp, int n_ops, int m, int nt)
{
    extern __shared__ double sig[];          // [4 waves][n_ops][2] of this workgroup's d
    for (int e = threadIdx.x; e < 4 * n_ops * 2; e += blockDim.x) sig[e] = 0.0;
    __syncthreads();
    DenseTile<DN_RB, DN_NG> t;
    const bool active = dense_tile(t, Np >> 4, Np >> 4, m, nt);      // "groups" = 16-row blocks r2 of psi
    const int d = t.sub;
    if (active) {
        const int PWc = 2 * cp;
        const size_t hstep = (size_t)Np * PWc;
        d4 aR[DN_RB][DN_NG], aA[DN_RB][DN_NG], aB[DN_RB][DN_NG];
        ZERO_ACC(aR); ZERO_ACC(aA); ZERO_ACC(aB);
        for (int i = 0; i + d < m; i++) {
            const int j = i + 1 + d;
            if (i > 0) {                     // running sum in units of the current weight 1/j: times (1/(j-1)) / (1/j)
                const double sc = (double)j / (double)(j - 1);
                #pragma unroll
                for (int r = 0; r < DN_RB; r++)
                    #pragma unroll
                    for (int g = 0; g < DN_NG; g++) { aR[r][g] *= sc; aA[r][g] *= sc; aB[r][g] *= sc; }
            }
            outer_any<M3>(aR, aA, aB, t, Gp + ((size_t)t.n * m + (j - 1)) * hstep,
                          (i == 0) ? hist + (size_t)t.n * hstep : dpsi + ((size_t)t.n * m + (i - 1)) * hstep, (size_t)PWc, cp >> 3);
        }
        outer_finish<M3>(aR, aA, aB);
        frobenius_sigma(aR, aA, t, ops, Np, n_ops, 1.0 / (double)m, wave_sig(sig, n_ops * 2));      // (the last pair of every d has j = m)
    }
    __syncthreads();
    if (!active && t.n >= nt) return;
    const int tile = t.slot - t.sub * t.ntile;       // plane = position of the tile inside its (n, d)
    const size_t plane = (size_t)nt * n_ops * m * 2;
    for (int e = threadIdx.x; e < n_ops * 2; e += blockDim.x)
        sigma[(size_t)tile * plane + (size_t)t.n * n_ops * m * 2 + ((e >> 1) * m + d) * 2 + (e & 1)] = sum_wave_sig(sig, n_ops * 2, e);
}

// ... and without the stage derivatives psi_i = D_i psi_0 at all:  M(i, j) = g_j psi_0^H D_i^H, so
//   stage 1 (k_gouter):   X_j = (1/j) g_j psi_0^H, j = 1..m            m products N x N x c, stored as panels [Np][2Np]
//   stage 2 (k_ginner_d): M_d = X_{d+1} + sum_{i>=1} X_{i+1+d} D_i^H     m(m-1)/2 products N x N x N with the D_i of the
//                                                                      step-matrix build (k_level_f) as they lie in HBM
// against m (k_derivs_f) + m(m+1)/2 (k_ginner_m) products N x N x c: fewer when (m-1) N < (m+1) c -- config 5 (c = N):
// 21 units instead of 27, and the 2.6 ms k_derivs_f launch disappears from the gradient evaluation.
template <int DN_RB, int DN_NG, bool M3>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_gouter(const double *__restrict__ hist, const double *__restrict__ Gp,
                                                double *__restrict__ X, int Np, int cp, int m, int nt, d2 *__restrict__ Xfrag)
{
    DenseTile<DN_RB, DN_NG> t;
    if (!dense_tile(t, Np >> 4, Np >> 4, m, nt)) return;
    const int j = t.sub + 1, PWc = 2 * cp, PW = 2 * Np;
    const size_t hstep = (size_t)Np * PWc, panel = (size_t)Np * PW;
    d4 aR[DN_RB][DN_NG], aA[DN_RB][DN_NG], aB[DN_RB][DN_NG];
    ZERO_ACC(aR); ZERO_ACC(aA); ZERO_ACC(aB);
    outer_any<M3>(aR, aA, aB, t, Gp + ((size_t)t.n * m + (j - 1)) * hstep, hist + (size_t)t.n * hstep, (size_t)PWc, cp >> 3);
    outer_finish<M3>(aR, aA, aB);
    double *out = X + ((size_t)t.n * m + (j - 1)) * panel;
    const double w = 1.0 / (double)j;
    #pragma unroll
    for (int r = 0; r < DN_RB; r++) {
        if (t.rb[r] < 0) continue;
        #pragma unroll
        for (int g = 0; g < DN_NG; g++) {
            if (t.g[g] < 0) continue;
            #pragma unroll
            for (int e = 0; e < 4; e++) {      // element (r1, r2 = g*16 + c16): column group r2 / 8, slot r2 % 8 (re), + 8 (im)
                double *o = out + (size_t)(t.rb[r] * 16 + t.kk + 4 * e) * PW + (t.g[g] * 2 + (t.c16 >> 3)) * 16 + (t.c16 & 7);
                o[0] = w * aR[r][g][e];
                o[8] = w * aA[r][g][e];
                if (Xfrag) Xfrag[((size_t)t.n * m + (j - 1)) * (size_t)Np * Np + frag_index(Np, t.rb[r] * 16 + t.kk + 4 * e, t.g[g] * 16 + t.c16)] =
                               (d2){w * aR[r][g][e], w * aA[r][g][e]};
            }
        }
    }
}

// WEIGHTED = false: X_j carries its weight 1/j (k_gouter).  WEIGHTED = true: the panels are the swept Y_j = g_j psi_0^H of the
// third form (below) and the weights 1/j are applied here, the running sum kept in units of the current pair's weight.
template <int DN_RB, int DN_NG, bool WEIGHTED, bool M3>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 3))) void k_ginner_d(const double *__restrict__ ops, const double *__restrict__ X,
                                                  const double *__restrict__ D, double *__restrict__ sigma, int Np, int n_ops, int m, int nt,
                                                  const d2 *__restrict__ Xfrag, const d2 *__restrict__ Dfrag)
{
    extern __shared__ double sig[];          // [4 waves][n_ops][2] of this workgroup's d
    for (int e = threadIdx.x; e < 4 * n_ops * 2; e += blockDim.x) sig[e] = 0.0;
    __syncthreads();
    DenseTile<DN_RB, DN_NG> t;
    const bool active = dense_tile(t, Np >> 4, Np >> 4, m, nt);
    const int d = t.sub;
    if (active) {
        const int PW = 2 * Np;
        const size_t panel = (size_t)Np * PW;
        const double *Xn = X + (size_t)t.n * m * panel, *Dn = D + (size_t)t.n * m * panel;     // X_j at slot j-1, D_i at slot i-1
        d4 aR[DN_RB][DN_NG], aA[DN_RB][DN_NG], aB[DN_RB][DN_NG];
        {   // i = 0: D_0 = I, the term is X_{d+1} itself  (M3: Re = P1 + P2, Im = P3 - P1 + P2 with P2 = 0)
            const double *x0 = Xn + (size_t)d * panel;
            #pragma unroll
            for (int r = 0; r < DN_RB; r++)
                #pragma unroll
                for (int g = 0; g < DN_NG; g++)
                    #pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const int rb = t.rb[r] >= 0 ? t.rb[r] : t.rb[0], gb = t.g[g] >= 0 ? t.g[g] : t.g[0];
                        const double *o = x0 + (size_t)(rb * 16 + t.kk + 4 * e) * PW + (gb * 2 + (t.c16 >> 3)) * 16 + (t.c16 & 7);
                        if (M3) { aR[r][g][e] = o[0]; aA[r][g][e] = 0.0; aB[r][g][e] = o[8] + o[0]; }
                        else { aR[r][g][e] = o[0]; aA[r][g][e] = o[8]; aB[r][g][e] = 0.0; }
                    }
        }
        for (int i = 1; i + d < m; i++) {
            if (WEIGHTED) {                  // (1/(j-1)) / (1/j), j = i + 1 + d
                const double sc = (double)(i + 1 + d) / (double)(i + d);
                #pragma unroll
                for (int r = 0; r < DN_RB; r++)
                    #pragma unroll
                    for (int g = 0; g < DN_NG; g++) { aR[r][g] *= sc; aA[r][g] *= sc; aB[r][g] *= sc; }
            }
            if (M3) outer_frag3(aR, aA, aB, t, Xfrag + ((size_t)t.n * m + (i + d)) * Np * Np, Dfrag + ((size_t)t.n * m + (i - 1)) * Np * Np, Np);
            else outer_tile(aR, aA, aB, t, Xn + (size_t)(i + d) * panel, Dn + (size_t)(i - 1) * panel, (size_t)PW, Np >> 3);
        }
        outer_finish<M3>(aR, aA, aB);
        frobenius_sigma(aR, aA, t, ops, Np, n_ops, WEIGHTED ? 1.0 / (double)m : 1.0, wave_sig(sig, n_ops * 2));
    }
    __syncthreads();
    if (!active && t.n >= nt) return;
    const int tile = t.slot - t.sub * t.ntile;       // plane = position of the tile inside its (n, d): one writer per entry
    const size_t plane = (size_t)nt * n_ops * m * 2;
    for (int e = threadIdx.x; e < n_ops * 2; e += blockDim.x)
        sigma[(size_t)tile * plane + (size_t)t.n * n_ops * m * 2 + ((e >> 1) * m + d) * 2 + (e & 1)] = sum_wave_sig(sig, n_ops * 2, e);
}

// Third form: the reverse sweep itself on N x N matrices.  g_j enters the scalars only through Y_j = g_j psi_0^H, the sweep
// g_i -= (1/j) A_{j-1-i} g_j is a multiplication from the LEFT, and every seed is a combination of two vectors,
// g_j = c_j dt^j lambda_{n+1} - c_j (-dt)^j lambda_n.  So: two outer products per time point (k_youter: Lambda+ =
// lambda_{n+1} psi_0^H, Lambda- = lambda_n psi_0^H), the m seeds Y_j = c_j dt^j Lambda+ - c_j (-dt)^j Lambda- (k_yinit,
// elementwise), the SAME sweep kernel on the Y panels (width N instead of c), then k_ginner_d with the weights 1/j.
// 2 + m(m-1)/2 + m(m-1)/2 units against m(m-1)/2 + m + m(m-1)/2 of the second form when c = N (config 5: 32 against 36).
template <int DN_RB, int DN_NG, bool M3>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_youter(const double *__restrict__ hist, const double *__restrict__ lam,
                                                double *__restrict__ T, int Np, int cp, int nt)
{
    DenseTile<DN_RB, DN_NG> t;
    if (!dense_tile(t, Np >> 4, Np >> 4, 2, nt)) return;
    const int PWc = 2 * cp, PW = 2 * Np;
    const size_t hstep = (size_t)Np * PWc, panel = (size_t)Np * PW;
    const int nl = t.sub == 0 ? t.n + 1 : t.n;          // Lambda+ pairs lambda_{n+1}, Lambda- pairs lambda_n, both with psi_0(t_n)
    if (nl < 1 || nl > nt - 1) return;                  // (k_yinit treats the missing one as zero)
    d4 aR[DN_RB][DN_NG], aA[DN_RB][DN_NG], aB[DN_RB][DN_NG];
    ZERO_ACC(aR); ZERO_ACC(aA); ZERO_ACC(aB);
    outer_any<M3>(aR, aA, aB, t, lam + (size_t)nl * hstep, hist + (size_t)t.n * hstep, (size_t)PWc, cp >> 3);
    outer_finish<M3>(aR, aA, aB);
    double *out = T + ((size_t)t.n * 2 + t.sub) * panel;
    #pragma unroll
    for (int r = 0; r < DN_RB; r++) {
        if (t.rb[r] < 0) continue;
        #pragma unroll
        for (int g = 0; g < DN_NG; g++) {
            if (t.g[g] < 0) continue;
            #pragma unroll
            for (int e = 0; e < 4; e++) {
                double *o = out + (size_t)(t.rb[r] * 16 + t.kk + 4 * e) * PW + (t.g[g] * 2 + (t.c16 >> 3)) * 16 + (t.c16 & 7);
                o[0] = aR[r][g][e];
                o[8] = aA[r][g][e];
            }
        }
    }
}

// Y[n][j-1] = c_j dt^j Lambda+[n] [n <= nt-2] - c_j (-dt)^j Lambda-[n] [n >= 1].  grid (ceil(panel/256), nt)
// only_last: just Y_m (right operand of the first sweep level); the other seeds are formed by that level's epilogue (k_gsweep_f3)
__global__ __launch_bounds__(256) void k_yinit(const double *__restrict__ T, const double *__restrict__ cw, double *__restrict__ Y,
                                               size_t panel, int m, int nt, double *__restrict__ Yfrag, int Np, int only_last)
{
    const int n = blockIdx.y;
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= panel) return;
    const double lx = (n <= nt - 2) ? T[((size_t)n * 2) * panel + e] : 0.0;
    const double ln = (n >= 1) ? T[((size_t)n * 2 + 1) * panel + e] : 0.0;
    for (int j = only_last ? m : 1; j <= m; j++) Y[((size_t)n * m + (j - 1)) * panel + e] = cw[2 * j] * lx - cw[2 * j + 1] * ln;
    if (Yfrag) {        // Y_m is never swept: its fragment-order copy here (the others: k_gsweep_f3)
        const int row = (int)(e / (2 * Np)), w = (int)(e % (2 * Np)), ccol = (w >> 4) * 8 + (w & 7), is_im = (w >> 3) & 1;
        Yfrag[((size_t)n * m + (m - 1)) * panel + 2 * frag_index(Np, row, ccol) + is_im] = cw[2 * m] * lx - cw[2 * m + 1] * ln;
    }
}

// which form the gradient scalars take on the N > 64 path: 0 operator applications (k_ginner_f), 1 outer products with the
// stage derivatives (k_ginner_m), 2 outer products through the stored D_i (k_gouter + k_ginner_d: no k_derivs_f).
// 3 the whole reverse sweep on the N x N matrices Y_j (k_youter + k_yinit + k_gsweep_f on Y + k_ginner_d: neither stage
// derivatives nor a sweep on the state panels).  QGD_PATHS=ginner=0|1|2|3 forces one (tests, A/B timing).
static int dense_sigma_form(const qgdk_ctx *c)
{
    if (const char *e = qgd_path("ginner")) { const int f = atoi(e); return (f >= 2 && !c->Xouter) ? 1 : f; }
    if (c->Np < 128 || c->cp < 64) return 0;
    const int m = c->m;
    if (c->Xouter && c->Tlam && (long long)(m * (m - 1) / 2) * c->Np < (long long)(m * (m + 1) / 2 - 2) * c->cp) return 3;
    if (c->Xouter && (long long)(m - 1) * c->Np < (long long)(m + 1) * c->cp && m * (m + 1) / 2 <= c->n_ops * m + m) return 2;
    return ((m + 1) / 2 < c->n_ops) ? 1 : 0;
}

// ---------------------------------------------------------------------------
// ONE step of a state chain as a GEMM over the whole chip:  out = P in  (ADJ: out = P^H in + forcing).  The chain kernels
// (qgd_k_chain.hip) keep a column tile of the state in LDS over all the steps of a block and so put (blocks x column
// tiles) workgroups on a chain; for the single sequential chains of the scan -- the states at the super-block starts,
// the prefix over the windows of the lower ranks: nblocks = 1, 32 workgroups at config 5, 28 us per dependent step --
// one launch per step with one 16 x 16 output tile per workgroup, its four waves splitting the contraction (256 workgroups
// at config 5, 48 MFMAs per wave), is the shorter path.
// ---------------------------------------------------------------------------
template <bool ADJ>
__global__ __launch_bounds__(256) void k_chain_step(const double *__restrict__ P, const double *__restrict__ in, double *__restrict__ out,
                                                    const double *__restrict__ forcing, int Np, int cp)
{
    // workgroup = one 16 x 16 tile of the output (row block, pair of column groups); its four waves split the contraction
    // (a quarter of K each: the dependent chain of loads is what a step costs, not its 192 MFMAs) and add up through LDS
    __shared__ double part[3][64][8];
    constexpr int DN_RB = 1, DN_NG = 2;
    const int nrb = Np >> 4, npair = (cp + 15) >> 4, ngroups = cp >> 3;
    const int rb = blockIdx.x / npair, pr = blockIdx.x % npair;
    if (rb >= nrb) return;
    DenseTile<1, 2> t;
    t.n = 0; t.sub = 0; t.rb[0] = rb; t.g[0] = 2 * pr; t.g[1] = (2 * pr + 1 < ngroups) ? 2 * pr + 1 : -1;
    t.lane = threadIdx.x & 63; t.c16 = t.lane & 15; t.kk = t.lane >> 4; t.sign_hi = 0;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int PWc = 2 * cp;
    // K = Np split in four ranges of whole 16-column chunks (Np is a multiple of 16)
    const int chunks = Np >> 4, c0 = (chunks * wave) >> 2, c1 = (chunks * (wave + 1)) >> 2;
    const int kbeg = c0 * 16, klen = (c1 - c0) * 16;
    d4 p1[1][1], p2[1][1], p3[1][1];
    ZERO_ACC3(p1); ZERO_ACC3(p2); ZERO_ACC3(p3);
    if (klen > 0) {
        const double *inb = in + (size_t)kbeg * PWc;
        if (ADJ) cgemm3_tile_panelH(p1, p2, p3, t, P + (size_t)kbeg * 2 * Np, inb, PWc, Np, klen);
        else cgemm3_tile_planes<false>(p1, p2, p3, t, P + (size_t)Np * kbeg, P + (size_t)Np * Np + (size_t)Np * kbeg, inb, PWc, Np, klen);
    }
    double v[8];
    #pragma unroll
    for (int e = 0; e < 4; e++) {
        if (ADJ) { v[2 * e] = p1[0][0][e] + p2[0][0][e]; v[2 * e + 1] = p3[0][0][e] - p1[0][0][e] + p2[0][0][e]; }
        else { v[2 * e] = p1[0][0][e] - p2[0][0][e]; v[2 * e + 1] = p3[0][0][e] - p1[0][0][e] - p2[0][0][e]; }
    }
    if (wave > 0) {
        #pragma unroll
        for (int q = 0; q < 8; q++) part[wave - 1][t.lane][q] = v[q];
    }
    __syncthreads();
    if (wave > 0) return;
    const int g = lane_group(t, 0);
    if (g < 0) return;
    #pragma unroll
    for (int e = 0; e < 4; e++) {
        const size_t o = (size_t)(rb * 16 + t.kk + 4 * e) * PWc + g * 16 + (t.c16 & 7);
        double vre = v[2 * e] + part[0][t.lane][2 * e] + part[1][t.lane][2 * e] + part[2][t.lane][2 * e];
        double vim = v[2 * e + 1] + part[0][t.lane][2 * e + 1] + part[1][t.lane][2 * e + 1] + part[2][t.lane][2 * e + 1];
        if (forcing) { vre += forcing[o]; vim += forcing[o + 8]; }
        out[o] = vre; out[o + 8] = vim;
    }
}

extern "C" {

int qgdk_dense_operator_frag(const qgdk_ctx *c)
{
    if (c->n_ops == 0) return 0;
    hipLaunchKernelGGL(k_operator_frag, dim3(c->Np * c->Np / 256, c->n_ops), dim3(256), 0, c->stream, c->ops,
                       reinterpret_cast<d2 *>(c->OpFrag), c->Np);
    return (int)hipGetLastError();
}

int qgdk_dense_build_LR(const qgdk_ctx *c)
{
    d2 *Af = reinterpret_cast<d2 *>(c->Afrag);
#define CALL_AF(N) hipLaunchKernelGGL((k_assemble_frag<N>), dim3(c->Np * c->Np / 256, c->m, c->nt), dim3(256), 0, c->stream, \
                                      c->ops, c->tab, Af, c->Np, c->n_ops, c->m)
    DISPATCH_NOPS(c->n_ops, CALL_AF)
#undef CALL_AF
    // Np/8 >= 9 column groups here.  (64-row tiles <1,4>, to have 6.3 rounds of workgroups instead of 3.14 at config 5,
    //  were slower: 7.95 vs 7.2 ms for the six levels.)
    const int grid = dense_grid(2, 4, c->Np / 16, c->Np / 8, 1, c->nt);
    for (int j = 0; j < c->m; j++)
        if (dense_3m())
            hipLaunchKernelGGL((k_level_f3<2, 4>), dim3(grid), dim3(256), 0, c->stream, Af, c->D, c->Dfrag, c->L, c->R, c->cw, c->Np, c->m,
                               c->nt, j, c->cw_host[2 * (j + 1) + 1], c->cw_host[2 * (j + 1)]);
        else
            hipLaunchKernelGGL((k_level_f<2, 4>), dim3(grid), dim3(256), 0, c->stream, Af, c->D, c->Dfrag, c->L, c->R, c->cw, c->Np, c->m,
                               c->nt, j, c->cw_host[2 * (j + 1) + 1], c->cw_host[2 * (j + 1)]);
    return (int)hipGetLastError();
}

// tile shape by the number of column groups of the state panels
#define DISPATCH_SHAPE(ngroups, CALL) do { if ((ngroups) >= 3) CALL(2, 4); else if ((ngroups) == 2) CALL(4, 2); else CALL(4, 1); } while (0)

// one chain step on the whole chip (see k_chain_step): P = column-major planes (forward) or the panel of P (adjoint, P^H)
int qgdk_dense_chain_step(hipStream_t stream, int adj, const double *P, const double *in, double *out, const double *forcing, int Np, int cp)
{
    const int grid = (Np / 16) * ((cp + 15) / 16);
    if (adj) hipLaunchKernelGGL((k_chain_step<true>), dim3(grid), dim3(256), 0, stream, P, in, out, forcing, Np, cp);
    else hipLaunchKernelGGL((k_chain_step<false>), dim3(grid), dim3(256), 0, stream, P, in, out, forcing, Np, cp);
    return (int)hipGetLastError();
}

size_t qgdk_dense_inverse_words(int Np, int nt)
{
    return (size_t)nt * (2 * (size_t)Np * 2 * Np + 2 * BINV_B * BINV_B) + (size_t)nt / 2 + 8;      // two panel buffers, D, flags
}

int qgdk_dense_inverse(const qgdk_ctx *c)
{
    if (!c->binv || !c->inv_scratch || !dense_3m() || qgd_path("binv_off") || c->nt < 2) return 0;
    const int Np = c->Np, nt = c->nt, nmat = nt - 1, PW = 2 * Np;
    const size_t panel = (size_t)Np * PW;
    double *PA = c->binv, *PB = PA + (size_t)nt * panel, *DkC = PB + (size_t)nt * panel;
    int *flags = reinterpret_cast<int *>(DkC + (size_t)nt * 2 * BINV_B * BINV_B);
    const double thresh = qgd_path("binv_thresh") ? atof(qgd_path("binv_thresh")) : 2.0;
    if (hipMemsetAsync(flags, 0, (size_t)nt * sizeof(int), c->stream) != hipSuccess) return -1;
    const double *Win = c->L;
    double *outs[2] = {PA, PB};
    int s = 0;
    for (int kb0 = 0; kb0 < Np; kb0 += BINV_B, s++) {
        const int bs = (Np - kb0 < BINV_B) ? Np - kb0 : BINV_B;
        double *Wout = outs[s & 1];
        hipLaunchKernelGGL(k_binv_planes, dim3(((Np + 31) / 32) * ((bs + 31) / 32) * nmat), dim3(256), 0, c->stream, Win, panel, c->LinvA,
                           (double *)nullptr, Np, kb0, bs);
        if (qgdk_inverse_diag(c, Win, panel, PW, (size_t)kb0 * PW + 2 * (size_t)kb0, bs, DkC, flags)) return -1;
        hipLaunchKernelGGL((k_binv_row<1, 4>), dim3(dense_grid(1, 4, bs / 16, Np / 8, 1, nmat)), dim3(256), 0, c->stream, DkC, Win, panel, Wout,
                           Np, nmat, kb0, bs);
        if (Np > bs)
            hipLaunchKernelGGL((k_binv_rest<2, 4>), dim3(dense_grid(2, 4, (Np - bs) / 16, Np / 8, 1, nmat)), dim3(256), 0, c->stream, c->LinvA,
                               Win, panel, Wout, Np, nmat, kb0, bs, flags, thresh);
        Win = Wout;
    }
    hipLaunchKernelGGL(k_binv_planes, dim3(((Np + 31) / 32) * ((Np + 31) / 32) * nmat), dim3(256), 0, c->stream, Win, panel, c->LinvA, c->LinvT,
                       Np, 0, Np);
    if (qgdk_inverse_redo(c, flags)) return -1;
    return hipGetLastError() == hipSuccess ? 1 : -1;
}

// returns 1 when the three-product propagator kernel took the launch (else the caller runs k_propagator)
int qgdk_dense_propagator(const qgdk_ctx *c)
{
    if (!dense_3m() || c->nt < 2) return 0;
    hipLaunchKernelGGL((k_propagator3<2, 4>), dim3(dense_grid(2, 4, c->Np / 16, c->Np / 8, 1, c->nt - 1)), dim3(256), 0, c->stream, c->LinvA,
                       c->R, c->Pr, c->Pc, c->Np, c->nt);
    return 1;
}

int qgdk_dense_derivs(const qgdk_ctx *c)
{
    const int ng = c->cp / 8;
#define CALL_DF(RB, NG) hipLaunchKernelGGL((k_derivs_f<RB, NG>), dim3(dense_grid(RB, NG, c->Np / 16, ng, c->m, c->nt, true)), dim3(256), 0, \
                                           c->stream, reinterpret_cast<const d2 *>(c->Dfrag), c->hist, c->dpsi, c->Np, c->cp, c->m, c->nt)
    if (dense_3m() && ng >= 3)
        hipLaunchKernelGGL((k_derivs_f3<2, 4>), dim3(dense_grid(2, 4, c->Np / 16, ng, c->m, c->nt, true)), dim3(256), 0, c->stream,
                           reinterpret_cast<const d2 *>(c->Dfrag), c->hist, c->dpsi, c->Np, c->cp, c->m, c->nt);
    else
        DISPATCH_SHAPE(ng, CALL_DF);
#undef CALL_DF
    return (int)hipGetLastError();
}

int qgdk_dense_lambda(const qgdk_ctx *c)
{
    const int ng = c->cp / 8;
#define CALL_LF(RB, NG) hipLaunchKernelGGL((k_lambda_f<RB, NG>), dim3(dense_grid(RB, NG, c->Np / 16, ng, 1, c->nt - 1)), dim3(256), 0, \
                                           c->stream, c->LinvT, c->yhist, c->lam, c->Np, c->cp, c->nt, c->sigma,                       \
                                           c->nt * c->n_ops * c->m * 2, c->grad, c->grad_accumulate ? 0 : c->n_pcof)
    if (dense_3m() && ng >= 3)
        hipLaunchKernelGGL((k_lambda_f3<2, 4>), dim3(dense_grid(2, 4, c->Np / 16, ng, 1, c->nt - 1)), dim3(256), 0, c->stream, c->LinvT, c->yhist,
                           c->lam, c->Np, c->cp, c->nt, c->sigma, c->nt * c->n_ops * c->m * 2, c->grad, c->grad_accumulate ? 0 : c->n_pcof);
    else
        DISPATCH_SHAPE(ng, CALL_LF);
#undef CALL_LF
    return (int)hipGetLastError();
}

// planes of sigma the form in use writes (one per tile of a time point that contributes to an entry; k_contract adds them in order)
static int sigma_planes_of(int form, int Np, int cp, int m)
{
    const int nrb = Np / 16, ng = cp / 8;
    if (form >= 1) return (((nrb + 1) / 2) * ((nrb + 1) / 2) + 3) / 4;               // <2,2> tiles over (row blocks, row blocks): workgroups per (n, d)
    const int RB = ng >= 3 ? 2 : 4, NG = ng >= 3 ? 4 : (ng == 2 ? 2 : 1);            // DISPATCH_SHAPE
    return (((nrb + RB - 1) / RB) * ((ng + NG - 1) / NG) * m + 3) / 4;        // (k_ginner_f: the source level belongs to the wave's item)
}
int qgdk_dense_sigma_planes(const qgdk_ctx *c) { return sigma_planes_of(dense_sigma_form(c), c->Np, c->cp, c->m); }
int qgdk_dense_sigma_form(const qgdk_ctx *c) { return dense_sigma_form(c); }      // (diagnostic: qgd_get_intermediate("selection"))
int qgdk_dense_sigma_planes_max(int Np, int cp, int m) { return std::max(sigma_planes_of(0, Np, cp, m), sigma_planes_of(1, Np, cp, m)); }

int qgdk_dense_gradient_needs_derivs(const qgdk_ctx *c) { return dense_sigma_form(c) < 2; }

int qgdk_dense_gradient(const qgdk_ctx *c)
{
    const size_t hstep = (size_t)c->Np * 2 * c->cp;
    const d2 *Af = reinterpret_cast<const d2 *>(c->Afrag);
    double *Gp = c->panel_scratch;
    const int ng = c->cp / 8;
    const bool m3 = dense_3m() && c->Xfrag;
    d2 *Xf = reinterpret_cast<d2 *>(c->Xfrag);
    const d2 *Df = reinterpret_cast<const d2 *>(c->Dfrag);
    if (dense_sigma_form(c) == 3) {      // the sweep on the matrices Y_j = g_j psi_0^H (see k_youter)
        const bool lazy = dense_3m() && c->m >= 2;      // seeds formed inside the first sweep level
        const size_t panel = (size_t)c->Np * 2 * c->Np;
        const int ogrid2 = dense_grid(2, 2, c->Np / 16, c->Np / 16, 2, c->nt), ogrid = dense_grid(2, 2, c->Np / 16, c->Np / 16, c->m, c->nt);
        if (dense_3m()) hipLaunchKernelGGL((k_youter<2, 2, true>), dim3(ogrid2), dim3(256), 0, c->stream, c->hist, c->lam, c->Tlam, c->Np, c->cp, c->nt);
        else hipLaunchKernelGGL((k_youter<2, 2, false>), dim3(ogrid2), dim3(256), 0, c->stream, c->hist, c->lam, c->Tlam, c->Np, c->cp, c->nt);
        hipLaunchKernelGGL(k_yinit, dim3((unsigned)((panel + 255) / 256), c->nt), dim3(256), 0, c->stream, c->Tlam, c->cw, c->Xouter, panel,
                           c->m, c->nt, m3 ? c->Xfrag : nullptr, c->Np, lazy ? 1 : 0);
        const int ngy = c->Np / 8;
        for (int j = c->m; j >= 2; j--) {
#define CALL_GY(RB, NG) hipLaunchKernelGGL((k_gsweep_f<RB, NG>), dim3(dense_grid(RB, NG, c->Np / 16, ngy, j - 1, c->nt, true)), dim3(256), 0, \
                                           c->stream, Af, c->Xouter, c->Np, c->Np, c->m, c->nt, j)
            if (dense_3m() && ngy >= 3)
                hipLaunchKernelGGL((k_gsweep_f3<2, 4>), dim3(dense_grid(2, 4, c->Np / 16, ngy, j - 1, c->nt, true)), dim3(256), 0, c->stream, Af,
                                   c->Xouter, c->Np, c->Np, c->m, c->nt, j, m3 ? Xf : nullptr, (lazy && j == c->m) ? c->Tlam : nullptr, c->cw);
            else
                DISPATCH_SHAPE(ngy, CALL_GY);
#undef CALL_GY
        }
        if (m3)
            hipLaunchKernelGGL((k_ginner_d<2, 2, true, true>), dim3(ogrid), dim3(256), (size_t)4 * c->n_ops * 2 * sizeof(double), c->stream, c->ops,
                               c->Xouter, c->D, c->sigma, c->Np, c->n_ops, c->m, c->nt, Xf, Df);
        else
            hipLaunchKernelGGL((k_ginner_d<2, 2, true, false>), dim3(ogrid), dim3(256), (size_t)4 * c->n_ops * 2 * sizeof(double), c->stream, c->ops,
                               c->Xouter, c->D, c->sigma, c->Np, c->n_ops, c->m, c->nt, Xf, Df);
        return (int)hipGetLastError();
    }
    hipLaunchKernelGGL(k_ginit, dim3((unsigned)((hstep + 255) / 256), c->nt), dim3(256), 0, c->stream, c->lam, c->cw, Gp, hstep,
                       c->m, c->nt);
    for (int j = c->m; j >= 2; j--) {
#define CALL_GS(RB, NG) hipLaunchKernelGGL((k_gsweep_f<RB, NG>), dim3(dense_grid(RB, NG, c->Np / 16, ng, j - 1, c->nt, true)), dim3(256), 0, \
                                           c->stream, Af, Gp, c->Np, c->cp, c->m, c->nt, j)
        if (dense_3m() && ng >= 3)
            hipLaunchKernelGGL((k_gsweep_f3<2, 4>), dim3(dense_grid(2, 4, c->Np / 16, ng, j - 1, c->nt, true)), dim3(256), 0, c->stream, Af, Gp,
                               c->Np, c->cp, c->m, c->nt, j, (d2 *)nullptr, (const double *)nullptr, c->cw);
        else
            DISPATCH_SHAPE(ng, CALL_GS);
#undef CALL_GS
    }
#define CALL_GI(RB, NG) hipLaunchKernelGGL((k_ginner_f<RB, NG>), dim3(dense_grid(RB, NG, c->Np / 16, ng, c->m, c->nt, true)), dim3(256), \
                                           (size_t)4 * c->n_ops * c->m * 2 * sizeof(double), c->stream,                         \
                                           reinterpret_cast<const d2 *>(c->OpFrag), c->hist, c->dpsi, Gp, c->sigma, c->Np, c->cp,  \
                                           c->n_ops, c->m, c->nt)
    // outer-product forms when they are fewer GEMM units and the contraction is long enough (dense_sigma_form)
    const int form = dense_sigma_form(c);
    const int ogrid = dense_grid(2, 2, c->Np / 16, c->Np / 16, c->m, c->nt);
    if (form == 2) {
        if (m3) {
            hipLaunchKernelGGL((k_gouter<2, 2, true>), dim3(ogrid), dim3(256), 0, c->stream, c->hist, Gp, c->Xouter, c->Np, c->cp, c->m, c->nt, Xf);
            hipLaunchKernelGGL((k_ginner_d<2, 2, false, true>), dim3(ogrid), dim3(256), (size_t)4 * c->n_ops * 2 * sizeof(double), c->stream, c->ops,
                               c->Xouter, c->D, c->sigma, c->Np, c->n_ops, c->m, c->nt, Xf, Df);
        } else {
            hipLaunchKernelGGL((k_gouter<2, 2, false>), dim3(ogrid), dim3(256), 0, c->stream, c->hist, Gp, c->Xouter, c->Np, c->cp, c->m, c->nt,
                               (d2 *)nullptr);
            hipLaunchKernelGGL((k_ginner_d<2, 2, false, false>), dim3(ogrid), dim3(256), (size_t)4 * c->n_ops * 2 * sizeof(double), c->stream, c->ops,
                               c->Xouter, c->D, c->sigma, c->Np, c->n_ops, c->m, c->nt, Xf, Df);
        }
        return (int)hipGetLastError();
    }
    if (form == 1) {
        if (dense_3m())
            hipLaunchKernelGGL((k_ginner_m<2, 2, true>), dim3(ogrid), dim3(256), (size_t)4 * c->n_ops * 2 * sizeof(double), c->stream, c->ops,
                               c->hist, c->dpsi, Gp, c->sigma, c->Np, c->cp, c->n_ops, c->m, c->nt);
        else
            hipLaunchKernelGGL((k_ginner_m<2, 2, false>), dim3(ogrid), dim3(256), (size_t)4 * c->n_ops * 2 * sizeof(double), c->stream, c->ops,
                               c->hist, c->dpsi, Gp, c->sigma, c->Np, c->cp, c->n_ops, c->m, c->nt);
        return (int)hipGetLastError();
    }
    DISPATCH_SHAPE(ng, CALL_GI);      // (a 2 x 2 tile at 4 waves per SIMD was slower: 13.2 vs 11.4 ms at config 5)
#undef CALL_GI
    return (int)hipGetLastError();
}

} // extern "C"
