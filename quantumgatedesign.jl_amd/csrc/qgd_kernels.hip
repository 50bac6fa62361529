// qgd_kernels.hip -- CDNA4 (gfx950) kernels of the time-parallel Hermite stepper
// and discrete adjoint.  See DESIGN.md for the algorithm, the data layout and
// the roofline of each kernel.  Reference behaviour being reproduced:
//   src/hermite.jl:56-101,389-427,556-588   (recursion, weights, Hamiltonian apply)
//   src/forward_evolution.jl:88-245,352-483 (forward / adjoint sweeps)
//   src/eval_grad_discrete_adjoint.jl:1-67,107-160,582-800
//   src/infidelity.jl:7-18,56-96
//
// Conventions
//   Real form w=[u;v] <-> psi=u+iv.  A=[K S;-S K] <-> K - iS (skew-Hermitian),
//   A^T <-> A^H = -A.
//   "panel": a complex [rows x C] block stored row-major as real [rows][2*Cp],
//   columns in groups of 16 = 8 real parts followed by the 8 imaginary parts of
//   the same 8 complex columns.  One 16x16 f64 MFMA output tile = 16 rows x 8
//   complex columns, and   C = Are*[Bre|Bim] + Aim*[-Bim|Bre]   needs no
//   cross-lane traffic.
//   "planes": a complex matrix as two real column-major Np x Np planes (re, im);
//   the natural layout of a LEFT operand (A fragment: 16 consecutive rows of one k).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "qgd_device.h"

typedef double d4 __attribute__((ext_vector_type(4)));

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return (int)e_; } while (0)
// hipFuncSetAttribute once per kernel instantiation (it is a host-side driver call)
#define SET_LDS_ONCE(fn, bytes) do { static size_t done_ = 0; if ((size_t)(bytes) > done_) { HIPCHK(hipFuncSetAttribute((const void *)(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes))); done_ = (bytes); } } while (0)

// B-operand pair for one 4-deep k-step out of a panel row (LDS or global):
// b1 = [Bre|Bim], b2 = [-Bim|Bre]
__device__ __forceinline__ void panel_b(const double *row16, int c16, double &b1, double &b2)
{
    b1 = row16[c16];
    double t = row16[c16 ^ 8];
    b2 = (c16 < 8) ? -t : t;
}

// ---------------------------------------------------------------------------
// A-fragment providers: value of the LEFT operand at (row, k)
// ---------------------------------------------------------------------------
struct OpCoef {           // coefficients of one derivative order d at one time point
    double sys;           // 1 if d == 0 else 0
    double p[QGD_MAX_OPS_DEV];
    double q[QGD_MAX_OPS_DEV];
};

// NOPS template parameter: the number of control operators at compile time (branch-free,
// all loads of one element issued together); NOPS = -1 keeps the count at run time.
#define NOPS_LIM(NOPS) ((NOPS) < 0 ? QGD_MAX_OPS_DEV : (NOPS))
#define NOPS_ON(NOPS, o, n_ops) ((NOPS) >= 0 || (o) < (n_ops))

// A_d(t_n)(row,k) = K_d - i S_d assembled from the fixed operators (hermite.jl:566-587)
template <int NOPS>
__device__ __forceinline__ void assembled_a(const double *__restrict__ ops, int Np, int n_ops,
                                            const OpCoef &cf, int row, int k, double &are, double &aim)
{
    const size_t e = (size_t)row + (size_t)Np * k;
    const size_t pl = (size_t)Np * Np;
    double K = cf.sys * ops[e];
    double S = cf.sys * ops[pl + e];
    #pragma unroll
    for (int o = 0; o < NOPS_LIM(NOPS); o++) {      // static indices: keeps cf in registers
        if (NOPS_ON(NOPS, o, n_ops)) {
            K += cf.q[o] * ops[(size_t)(2 + 2 * o) * pl + e];
            S += cf.p[o] * ops[(size_t)(3 + 2 * o) * pl + e];
        }
    }
    are = K;
    aim = -S;
}

__device__ __forceinline__ void load_coef(OpCoef &cf, const double *__restrict__ tab, int n, int d,
                                          int m, int n_ops)
{   // tab[n][d][k][2], d = 0..m
    cf.sys = (d == 0) ? 1.0 : 0.0;
    const double *t = tab + (((size_t)n * (m + 1) + d) * n_ops) * 2;
    #pragma unroll
    for (int o = 0; o < QGD_MAX_OPS_DEV; o++) {
        cf.p[o] = (o < n_ops) ? t[2 * o] : 0.0;
        cf.q[o] = (o < n_ops) ? t[2 * o + 1] : 0.0;
    }
}

// ---------------------------------------------------------------------------
// K0: control tables  tab[n][d][k][pq] = sum_l G[k][n][d][l] * pcof[off_k + l]
// (fill_p_mat!/fill_q_mat!, Control.jl:125-149, for the whole grid at once)
// ---------------------------------------------------------------------------
__global__ void k_tables(const double *__restrict__ G, const int64_t *__restrict__ goff,
                         const int32_t *__restrict__ ncoef, const int32_t *__restrict__ poff,
                         const double *__restrict__ pcof, double *__restrict__ tab, int nt, int m,
                         int n_ops, double *__restrict__ scal, int *__restrict__ status)
{
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < 4) scal[idx] = 0.0;            // objective scalars and the singularity flag start at zero
    if (idx == 4) *status = 0;
    int total = nt * (m + 1) * n_ops * 2;
    if (idx >= total) return;
    int pq = idx & 1;
    int k = (idx >> 1) % n_ops;
    int d = ((idx >> 1) / n_ops) % (m + 1);
    int n = ((idx >> 1) / n_ops) / (m + 1);
    int nc = ncoef[k];
    // G for control k: [pq][nt][m+1][nc]
    const double *g = G + goff[k] + (((size_t)pq * nt + n) * (m + 1) + d) * nc;
    const double *pc = pcof + poff[k];
    double s = 0.0;
    for (int l = 0; l < nc; l++) s += g[l] * pc[l];
    tab[idx] = s;
}

// general path: tables given by the host in Julia layout [(1+m), n_ops, nt]
__global__ void k_tables_from_host(const double *__restrict__ pt, const double *__restrict__ qt,
                                   double *__restrict__ tab, int nt, int m, int n_ops)
{
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    int total = nt * (m + 1) * n_ops * 2;
    if (idx >= total) return;
    int pq = idx & 1;
    int k = (idx >> 1) % n_ops;
    int d = ((idx >> 1) / n_ops) % (m + 1);
    int n = ((idx >> 1) / n_ops) / (m + 1);
    const double *src = pq ? qt : pt;
    tab[idx] = src[d + (size_t)(m + 1) * (k + (size_t)n_ops * n)];
}

// ---------------------------------------------------------------------------
// K1: one level of the Taylor-coefficient recursion on the identity
//   D_{j+1}(t_n) = 1/(j+1) * ( sum_{i=1..j} A_{j-i}(t_n) D_i(t_n) + A_j(t_n) )
// (compute_derivatives! hermite.jl:56-101 applied to every unit vector = form_LHS/
// form_RHS hermite.jl:594-640), fused with the Hermite weights
//   L += c_{j+1} (-dt)^{j+1} D_{j+1},  R += c_{j+1} dt^{j+1} D_{j+1}   (hermite.jl:394-427)
// Block: 256 threads = 4 waves; tile 64 rows x NG col-groups; grid (tiles, nt).
// D: [nt][m][Np][2Np] panels.
// ---------------------------------------------------------------------------
#define LV_NG 4
#define LV_KC 16
template <int NOPS>
__global__ __launch_bounds__(256) void k_level(const double *__restrict__ ops,
                                               const double *__restrict__ tab,
                                               double *__restrict__ D, double *__restrict__ L,
                                               double *__restrict__ R, int Np, int n_ops, int m,
                                               int j, double cL, double cR)
{
    __shared__ __attribute__((aligned(32))) double Bs[LV_KC][16 * LV_NG];
    const int n = blockIdx.y;
    const int ngroups = Np / 8;
    const int gtiles = (ngroups + LV_NG - 1) / LV_NG;
    const int gb = blockIdx.x % gtiles, rb4 = blockIdx.x / gtiles;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c16 = lane & 15, kk = lane >> 4;
    const int rb = rb4 * 4 + wave;               // 16-row block of this wave
    const bool row_ok = rb * 16 < Np;
    const int arow = rb * 16 + c16;              // row this lane feeds as A operand
    const int PW = 2 * Np;
    const size_t panel = (size_t)Np * PW;
    double *Dn = D + (size_t)n * m * panel;

    d4 acc[LV_NG];
    for (int g = 0; g < LV_NG; g++) acc[g] = (d4){0, 0, 0, 0};

    for (int i = 1; i <= j; i++) {
        OpCoef cf;
        load_coef(cf, tab, n, j - i, m, n_ops);
        const double *Bsrc = Dn + (size_t)(i - 1) * panel;   // D_i
        for (int kc = 0; kc < Np; kc += LV_KC) {
            // stage B tile: rows kc..kc+15, cols gb*64 .. +63
            {
                int r = threadIdx.x >> 4, c4 = (threadIdx.x & 15) * 4;
                int gcol = gb * 16 * LV_NG + c4;
                double4 v = make_double4(0, 0, 0, 0);
                if (gcol < PW) v = *reinterpret_cast<const double4 *>(Bsrc + (size_t)(kc + r) * PW + gcol);
                *reinterpret_cast<double4 *>(&Bs[r][c4]) = v;
            }
            double are[4], aim[4];
            if (row_ok) {
                #pragma unroll
                for (int s = 0; s < 4; s++) assembled_a<NOPS>(ops, Np, n_ops, cf, arow, kc + 4 * s + kk, are[s], aim[s]);
            }
            __syncthreads();
            if (row_ok) {
                #pragma unroll
                for (int s = 0; s < 4; s++) {
                    #pragma unroll
                    for (int g = 0; g < LV_NG; g++) {
                        double b1, b2;
                        panel_b(&Bs[4 * s + kk][16 * g], c16, b1, b2);
                        acc[g] = MFMA(are[s], b1, acc[g]);
                        acc[g] = MFMA(aim[s], b2, acc[g]);
                    }
                }
            }
            __syncthreads();
        }
    }
    if (!row_ok) return;
    // epilogue: + A_j, scale, store D_{j+1}, accumulate L and R
    OpCoef cj;
    load_coef(cj, tab, n, j, m, n_ops);
    const double inv = 1.0 / (double)(j + 1);
    const size_t pl = (size_t)Np * Np;
    double *Dout = Dn + (size_t)j * panel;
    double *Ln = L + (size_t)n * panel, *Rn = R + (size_t)n * panel;
    #pragma unroll
    for (int g = 0; g < LV_NG; g++) {
        const int grp = gb * LV_NG + g;
        if (grp >= ngroups) continue;
        const int ccol = grp * 8 + (c16 & 7);        // complex column
        const bool is_im = c16 >= 8;
        #pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = rb * 16 + kk + 4 * r;
            // A_j(row, ccol): read the transposed element so that lanes are contiguous:
            // K antisymmetric, S symmetric (SchrodingerProb.jl:73-101)
            const size_t e = (size_t)ccol + (size_t)Np * row;
            double add;
            if (!is_im) {
                double K = cj.sys * ops[e];
                #pragma unroll
                for (int o = 0; o < NOPS_LIM(NOPS); o++) if (NOPS_ON(NOPS, o, n_ops)) K += cj.q[o] * ops[(size_t)(2 + 2 * o) * pl + e];
                add = -K;                              // K(row,ccol) = -K(ccol,row)
            } else {
                double S = cj.sys * ops[pl + e];
                #pragma unroll
                for (int o = 0; o < NOPS_LIM(NOPS); o++) if (NOPS_ON(NOPS, o, n_ops)) S += cj.p[o] * ops[(size_t)(3 + 2 * o) * pl + e];
                add = -S;                              // Im A = -S
            }
            const double val = (acc[g][r] + add) * inv;
            const size_t o = (size_t)row * PW + grp * 16 + c16;
            Dout[o] = val;
            if (j == 0) {
                const double id = (!is_im && row == ccol) ? 1.0 : 0.0;
                Ln[o] = id + cL * val;
                Rn[o] = id + cR * val;
            } else {
                Ln[o] += cL * val;
                Rn[o] += cR * val;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// K1 (fast path, Np = 64): all levels of the recursion fused into one launch.
// One workgroup = (time point, half of the columns): 512 threads = 8 waves,
// wave = (16-row block, pair of column groups).  D_1..D_{m-1} of the workgroup's 32
// complex columns stay in LDS (column slabs of the recursion are independent), L and R
// stay in registers, and the work is ordered by SOURCE: when D_i is complete its
// contributions A_d D_i to every later level are accumulated at once, so each operator
// element is fetched once per (source, k) and serves up to m-1 MFMA pairs.
// No global round trips between levels, m-1 barriers per workgroup.
// ---------------------------------------------------------------------------
struct OpVals { double K0, S0, K[QGD_MAX_OPS_DEV], S[QGD_MAX_OPS_DEV]; };

template <int NOPS>
__device__ __forceinline__ void load_opvals(OpVals &v, const double *__restrict__ ops, int Np, int n_ops, size_t e)
{
    const size_t pl = (size_t)Np * Np;
    v.K0 = ops[e];
    v.S0 = ops[pl + e];
    #pragma unroll
    for (int o = 0; o < NOPS_LIM(NOPS); o++) {
        if (NOPS_ON(NOPS, o, n_ops)) { v.K[o] = ops[(size_t)(2 + 2 * o) * pl + e]; v.S[o] = ops[(size_t)(3 + 2 * o) * pl + e]; }
        else { v.K[o] = 0.0; v.S[o] = 0.0; }
    }
}

// coefficient block in LDS: cf[d][0] = sys flag, cf[d][1+2o] = p, cf[d][2+2o] = q
#define CF_STRIDE (1 + 2 * QGD_MAX_OPS_DEV)
template <int NOPS>
__device__ __forceinline__ void combine_opvals(const OpVals &v, const double *cf, int n_ops, double &are, double &aim)
{
    double K = cf[0] * v.K0, S = cf[0] * v.S0;
    #pragma unroll
    for (int o = 0; o < NOPS_LIM(NOPS); o++)
        if (NOPS_ON(NOPS, o, n_ops)) { K += cf[2 + 2 * o] * v.K[o]; S += cf[1 + 2 * o] * v.S[o]; }
    are = K;
    aim = -S;
}

template <int M, int NOPS>
__global__ __launch_bounds__(512) void k_build_LR64(const double *__restrict__ ops,
                                                    const double *__restrict__ tab,
                                                    double *__restrict__ L, double *__restrict__ R,
                                                    int n_ops, const double *__restrict__ cw)
{
    constexpr int NP = 64, NGW = 4, SW = 16 * NGW;   // slab width in doubles
    constexpr int ND = (M > 1) ? M - 1 : 1;
    extern __shared__ double smem[];
    double *Dbuf = smem;                               // [M-1][NP][SW]
    const int n = blockIdx.x >> 1, h = blockIdx.x & 1;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int c16 = lane & 15, kk = lane >> 4;
    const int rb = wave & 3, gh = wave >> 2;           // row block, pair of groups inside the slab
    const int arow = rb * 16 + c16;
    constexpr int PW = 2 * NP;
    const size_t panel = (size_t)NP * PW, pl = (size_t)NP * NP;

    d4 Lacc[2], Racc[2], T[M][2];                      // T[q]: accumulator of D_{q+1}
    #pragma unroll
    for (int g = 0; g < 2; g++) {
        #pragma unroll
        for (int q = 0; q < M; q++) T[q][g] = (d4){0, 0, 0, 0};
        const int ccol = (h * NGW + gh * 2 + g) * 8 + (c16 & 7);
        #pragma unroll
        for (int r = 0; r < 4; r++) {
            const double id = (c16 < 8 && rb * 16 + kk + 4 * r == ccol) ? 1.0 : 0.0;
            Lacc[g][r] = id;
            Racc[g][r] = id;
        }
    }

    #pragma unroll
    for (int i = 0; i < M; i++) {                      // source D_i (D_0 = I) feeds levels i+1 .. M
        const double *Dsrc = Dbuf + (size_t)(i > 0 ? i - 1 : 0) * NP * SW;
        // the identity slab is non-zero only for k inside the slab's own 32 complex columns
        const int ks0 = (i == 0) ? h * 8 : 0;
        // coefficients of the derivative orders this source needs, in registers
        // (uniform addresses: the compiler keeps them in scalar registers)
        double cfr[M][CF_STRIDE];
        #pragma unroll
        for (int d = 0; d + i < M; d++) {
            cfr[d][0] = (d == 0) ? 1.0 : 0.0;
            #pragma unroll
            for (int o = 0; o < NOPS_LIM(NOPS); o++) {
                const bool on = NOPS_ON(NOPS, o, n_ops);
                cfr[d][1 + 2 * o] = on ? tab[(((size_t)n * (M + 1) + d) * n_ops + o) * 2] : 0.0;
                cfr[d][2 + 2 * o] = on ? tab[(((size_t)n * (M + 1) + d) * n_ops + o) * 2 + 1] : 0.0;
            }
        }
        // software pipeline: operator elements 2 k-steps ahead (ring of 3), B fragments 1 ahead
        const int NKS = (i == 0) ? 8 : NP / 4;      // constant once the source loop is unrolled
        OpVals ring[3];
        #pragma unroll
        for (int q = 0; q < 2; q++)
            load_opvals<NOPS>(ring[q], ops, NP, n_ops, (size_t)arow + (size_t)NP * ((ks0 + q) * 4 + kk));
        double b1[2][2], b2[2][2];                     // [buffer][group]
        auto load_b = [&](int buf, int k) {
            #pragma unroll
            for (int g = 0; g < 2; g++) {
                if (i == 0) {
                    const int ccol = (h * NGW + gh * 2 + g) * 8 + (c16 & 7);
                    const double one = (k == ccol) ? 1.0 : 0.0;
                    b1[buf][g] = (c16 < 8) ? one : 0.0;  // [Bre|Bim] of the identity
                    b2[buf][g] = (c16 < 8) ? 0.0 : one;  // [-Bim|Bre]
                } else {
                    panel_b(Dsrc + (size_t)k * SW + (gh * 2 + g) * 16, c16, b1[buf][g], b2[buf][g]);
                }
            }
        };
        load_b(0, ks0 * 4 + kk);
        #pragma unroll
        for (int q = 0; q < NKS; q++) {
            const int k = (ks0 + q) * 4 + kk;
            if (q + 2 < NKS) load_opvals<NOPS>(ring[(q + 2) % 3], ops, NP, n_ops, (size_t)arow + (size_t)NP * (k + 8));
            if (q + 1 < NKS) load_b((q + 1) & 1, k + 4);
            #pragma unroll
            for (int d = 0; d + i < M; d++) {          // target level i+d+1
                double are, aim;
                combine_opvals<NOPS>(ring[q % 3], cfr[d], n_ops, are, aim);
                #pragma unroll
                for (int g = 0; g < 2; g++) {
                    T[i + d][g] = MFMA(are, b1[q & 1][g], T[i + d][g]);
                    T[i + d][g] = MFMA(aim, b2[q & 1][g], T[i + d][g]);
                }
            }
        }
        // D_{i+1} = T[i]/(i+1)
        const double inv = 1.0 / (double)(i + 1);
        const double cL = cw[2 * (i + 1) + 1], cR = cw[2 * (i + 1)];
        #pragma unroll
        for (int g = 0; g < 2; g++) {
            const int gl = gh * 2 + g;
            #pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = rb * 16 + kk + 4 * r;
                const double val = T[i][g][r] * inv;
                if (i + 1 < M) Dbuf[(size_t)i * NP * SW + (size_t)row * SW + gl * 16 + c16] = val;
                Lacc[g][r] += cL * val;
                Racc[g][r] += cR * val;
            }
        }
        if (i + 1 < M) __syncthreads();
    }
    double *Ln = L + (size_t)n * panel, *Rn = R + (size_t)n * panel;
    #pragma unroll
    for (int g = 0; g < 2; g++) {
        const int grp = h * NGW + gh * 2 + g;
        #pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = rb * 16 + kk + 4 * r;
            Ln[(size_t)row * PW + grp * 16 + c16] = Lacc[g][r];
            Rn[(size_t)row * PW + grp * 16 + c16] = Racc[g][r];
        }
    }
}

#define DISPATCH_NOPS(n_ops, CALL) \
    switch (n_ops) { case 1: CALL(1); break; case 2: CALL(2); break; case 3: CALL(3); break; case 4: CALL(4); break; \
                     default: CALL(-1); break; }

template <int M, int NOPS>
static int launch_build_LR64_n(const qgdk_ctx *c)
{
    const size_t shm = ((size_t)((M > 1) ? M - 1 : 1) * 64 * 64 + (size_t)M * CF_STRIDE) * sizeof(double);
    hipError_t e = hipFuncSetAttribute((const void *)k_build_LR64<M, NOPS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((k_build_LR64<M, NOPS>), dim3(2 * c->nt), dim3(512), shm, c->stream, c->ops, c->tab, c->L, c->R,
                       c->n_ops, c->cw);
    return (int)hipGetLastError();
}

template <int M>
static int launch_build_LR64(const qgdk_ctx *c)
{
#define CALL_LR(N) return launch_build_LR64_n<M, N>(c)
    DISPATCH_NOPS(c->n_ops, CALL_LR)
#undef CALL_LR
    return 0;
}

// ---------------------------------------------------------------------------
// K2: batched complex inverse by in-place Gauss-Jordan with partial pivoting.
// One workgroup per matrix.  Input L[n] (panel).  Outputs:
//   LinvA[n]: planes, column-major  (left operand of P = Linv * R)
//   LinvT[n]: planes, row-major     (left operand of lambda = Linv^H y)
// The work matrix lives in LDS when it fits, otherwise in a global scratch slab.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_inverse(const double *__restrict__ L,
                                                 double *__restrict__ LinvA,
                                                 double *__restrict__ LinvT,
                                                 double *__restrict__ scratch, int Np, int n0,
                                                 int use_lds, int *__restrict__ status)
{
    extern __shared__ double smem[];
    const int n = n0 + blockIdx.x;
    const int t = threadIdx.x, nth = blockDim.x;
    const int PW = 2 * Np;
    const size_t panel = (size_t)Np * PW, pl = (size_t)Np * Np;
    // aux (always LDS): colre[Np], colim[Np], perm[Np] (as double), red[2*nth/64...]
    double *aux = smem;
    double *colre = aux, *colim = aux + Np;
    int *perm = reinterpret_cast<int *>(aux + 2 * Np);
    double *redv = aux + 3 * Np;              // [4] values
    int *redi = reinterpret_cast<int *>(redv + 8);   // [4] indices
    double *M = use_lds ? (aux + 3 * Np + 16) : (scratch + (size_t)blockIdx.x * 2 * pl);
    double *Mre = M, *Mim = M + pl;           // row-major: (r,c) at r*Np + c

    const double *Ln = L + (size_t)n * panel;
    for (size_t e = t; e < pl; e += nth) {
        int r = e / Np, c = e % Np;
        Mre[e] = Ln[(size_t)r * PW + (c >> 3) * 16 + (c & 7)];
        Mim[e] = Ln[(size_t)r * PW + (c >> 3) * 16 + 8 + (c & 7)];
    }
    __syncthreads();

    for (int p = 0; p < Np; p++) {
        // pivot search over rows p..Np-1 of column p
        double best = -1.0; int bi = p;
        for (int r = p + t; r < Np; r += nth) {
            double a = Mre[(size_t)r * Np + p], b = Mim[(size_t)r * Np + p];
            double v = a * a + b * b;
            if (v > best) { best = v; bi = r; }
        }
        for (int off = 32; off > 0; off >>= 1) {
            double ob = __shfl_down(best, off);
            int oi = __shfl_down(bi, off);
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        if ((t & 63) == 0) { redv[t >> 6] = best; redi[t >> 6] = bi; }
        __syncthreads();
        int pr = redi[0]; double pb = redv[0];
        for (int w = 1; w < (nth >> 6); w++)
            if (redv[w] > pb || (redv[w] == pb && redi[w] < pr)) { pb = redv[w]; pr = redi[w]; }
        if (t == 0) { perm[p] = pr; if (!(pb > 0.0)) *status = 1; }
        // swap rows p and pr
        if (pr != p) {
            for (int c = t; c < Np; c += nth) {
                double a = Mre[(size_t)p * Np + c]; Mre[(size_t)p * Np + c] = Mre[(size_t)pr * Np + c]; Mre[(size_t)pr * Np + c] = a;
                double b = Mim[(size_t)p * Np + c]; Mim[(size_t)p * Np + c] = Mim[(size_t)pr * Np + c]; Mim[(size_t)pr * Np + c] = b;
            }
        }
        __syncthreads();
        // save column p, then a_pp <- 1, a_ip <- 0
        for (int r = t; r < Np; r += nth) {
            colre[r] = Mre[(size_t)r * Np + p];
            colim[r] = Mim[(size_t)r * Np + p];
        }
        __syncthreads();
        for (int r = t; r < Np; r += nth) {
            Mre[(size_t)r * Np + p] = (r == p) ? 1.0 : 0.0;
            Mim[(size_t)r * Np + p] = 0.0;
        }
        __syncthreads();
        // scale pivot row by 1/pivot
        {
            double a = colre[p], b = colim[p];
            double den = 1.0 / (a * a + b * b);
            double ir = a * den, ii = -b * den;
            for (int c = t; c < Np; c += nth) {
                double x = Mre[(size_t)p * Np + c], y = Mim[(size_t)p * Np + c];
                Mre[(size_t)p * Np + c] = x * ir - y * ii;
                Mim[(size_t)p * Np + c] = x * ii + y * ir;
            }
        }
        __syncthreads();
        // eliminate: row_i -= col_i * row_p  (i != p)
        for (size_t e = t; e < pl; e += nth) {
            int r = e / Np, c = e % Np;
            if (r == p) continue;
            double fr = colre[r], fi = colim[r];
            double x = Mre[(size_t)p * Np + c], y = Mim[(size_t)p * Np + c];
            Mre[e] -= fr * x - fi * y;
            Mim[e] -= fr * y + fi * x;
        }
        __syncthreads();
    }
    // undo the row swaps as column swaps in reverse order
    for (int p = Np - 1; p >= 0; p--) {
        int pr = perm[p];
        if (pr != p) {
            for (int r = t; r < Np; r += nth) {
                double a = Mre[(size_t)r * Np + p]; Mre[(size_t)r * Np + p] = Mre[(size_t)r * Np + pr]; Mre[(size_t)r * Np + pr] = a;
                double b = Mim[(size_t)r * Np + p]; Mim[(size_t)r * Np + p] = Mim[(size_t)r * Np + pr]; Mim[(size_t)r * Np + pr] = b;
            }
            __syncthreads();
        }
    }
    double *A = LinvA + (size_t)n * 2 * pl, *T = LinvT + (size_t)n * 2 * pl;
    for (size_t e = t; e < pl; e += nth) {       // row-major planes: straight copy
        T[e] = Mre[e];
        T[pl + e] = Mim[e];
    }
    for (size_t e = t; e < pl; e += nth) {       // column-major planes
        int c = e / Np, r = e % Np;
        A[e] = Mre[(size_t)r * Np + c];
        A[pl + e] = Mim[(size_t)r * Np + c];
    }
}

// ---------------------------------------------------------------------------
// K2 (fast path, Np <= 64): the same Gauss-Jordan inverse with the matrix held in
// registers: 256 threads as a 16x16 grid, thread (ty,tx) owns the BSxBS block of rows
// ty*BS.. and columns tx*BS.. (BS = Np/16).  Per pivot only the pivot column, the pivot
// row and the swapped row travel through LDS (2 barriers); the rank-1 update is local.
// ---------------------------------------------------------------------------
template <int NP, int TG>
__global__ __launch_bounds__(TG * TG) void k_inverse_reg(const double *__restrict__ L,
                                                     double *__restrict__ LinvA,
                                                     double *__restrict__ LinvT, int n0,
                                                     int *__restrict__ status)
{
    constexpr int BS = NP / TG, NTH = TG * TG;   // TG x TG threads, BS x BS block each
    static_assert(BS * TG == NP && BS <= 8, "block");
    __shared__ double colre[2][NP], colim[2][NP];
    __shared__ double rowre[2][2][NP], rowim[2][2][NP];
    __shared__ int perm[NP], outpos[NP], idx[NP];
    __shared__ double pivinv[2][2];
    const int n = n0 + blockIdx.x;
    const int t = threadIdx.x, ty = t / TG, tx = t % TG, lane = t & 63;
    constexpr int PW = 2 * NP;
    const size_t panel = (size_t)NP * PW, pl = (size_t)NP * NP;
    const double *Ln = L + (size_t)n * panel;

    double are[BS][BS], aim[BS][BS];
    #pragma unroll
    for (int i = 0; i < BS; i++)
        #pragma unroll
        for (int j = 0; j < BS; j++) {
            const int r = ty * BS + i, c = tx * BS + j;
            are[i][j] = Ln[(size_t)r * PW + (c >> 3) * 16 + (c & 7)];
            aim[i][j] = Ln[(size_t)r * PW + (c >> 3) * 16 + 8 + (c & 7)];
        }

    // statically indexed access to row / column `ii` of the register block (ii is uniform)
#define INV_SEL(ii, STMT) switch (ii) { \
    case 0: { constexpr int I = 0; STMT } break; \
    case 1: { constexpr int I = (1 < BS) ? 1 : 0; STMT } break; \
    case 2: { constexpr int I = (2 < BS) ? 2 : 0; STMT } break; \
    case 3: { constexpr int I = (3 < BS) ? 3 : 0; STMT } break; \
    case 4: { constexpr int I = (4 < BS) ? 4 : 0; STMT } break; \
    case 5: { constexpr int I = (5 < BS) ? 5 : 0; STMT } break; \
    case 6: { constexpr int I = (6 < BS) ? 6 : 0; STMT } break; \
    default: { constexpr int I = (7 < BS) ? 7 : 0; STMT } break; }
    for (int p = 0; p < NP; p++) {
        const int buf = p & 1;
        const int pblk = p / BS, poff = p % BS;        // uniform
        if (tx == pblk) {                               // publish column p
            INV_SEL(poff, _Pragma("unroll") for (int i = 0; i < BS; i++) {
                colre[buf][ty * BS + i] = are[i][I]; colim[buf][ty * BS + i] = aim[i][I]; })
        }
        if (ty == pblk) {                               // publish row p (before any swap)
            INV_SEL(poff, _Pragma("unroll") for (int j = 0; j < BS; j++) {
                rowre[buf][0][tx * BS + j] = are[I][j]; rowim[buf][0][tx * BS + j] = aim[I][j]; })
        }
        __syncthreads();
        // pivot search, redundantly in every wave.  One 64-bit key per lane: the bit pattern of
        // |a|^2 (non-negative doubles order like unsigned integers) with the low 6 mantissa bits
        // replaced by 63-row, so that one max-reduction yields the arg-max (ties -> lowest row).
        unsigned long long key = 0;
        if (lane < NP && lane >= p) {
            const double a = colre[buf][lane], b = colim[buf][lane];
            key = ((unsigned long long)__double_as_longlong(a * a + b * b) & ~63ull) | (unsigned long long)(63 - lane);
        }
        #pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned long long ok = __shfl_xor(key, off);
            key = ok > key ? ok : key;
        }
        const int pr = __builtin_amdgcn_readfirstlane(63 - (int)(key & 63ull));
        if (t == 0) { perm[p] = pr; if (!((key >> 6) != 0)) *status = 1; }
        const int rblk = pr / BS, roff = pr % BS;
        if (ty == rblk) {                               // owners of row pr publish it scaled by 1/pivot
            const double pa = colre[buf][pr], pb = colim[buf][pr];
            const double den = 1.0 / (pa * pa + pb * pb);
            const double ir = pa * den, ii = -pb * den;
            INV_SEL(roff, _Pragma("unroll") for (int j = 0; j < BS; j++) {
                const double x = are[I][j]; const double y = aim[I][j];
                rowre[buf][1][tx * BS + j] = x * ir - y * ii; rowim[buf][1][tx * BS + j] = x * ii + y * ir; })
            if (tx == 0) { pivinv[buf][0] = ir; pivinv[buf][1] = ii; }
        }
        const double cpr = colre[buf][p], cpi = colim[buf][p];   // old a_pp: multiplier of the swapped row
        __syncthreads();
        const double ir = pivinv[buf][0], ii = pivinv[buf][1];  // 1/pivot
        if (ty == rblk && pr != p) {                    // row swap: row pr takes the old row p
            INV_SEL(roff, _Pragma("unroll") for (int j = 0; j < BS; j++) {
                are[I][j] = rowre[buf][0][tx * BS + j]; aim[I][j] = rowim[buf][0][tx * BS + j]; })
        }
        double rpr[BS], rpi[BS], ur[BS], ui[BS];        // scaled pivot row; the same with column p zeroed
        #pragma unroll
        for (int j = 0; j < BS; j++) {
            const int c = tx * BS + j;
            rpr[j] = rowre[buf][1][c];
            rpi[j] = rowim[buf][1][c];
            ur[j] = (c == p) ? 0.0 : rpr[j];
            ui[j] = (c == p) ? 0.0 : rpi[j];
        }
        double fr[BS], fi[BS];                          // multipliers of my rows (0 for the pivot row)
        #pragma unroll
        for (int i = 0; i < BS; i++) {
            const int r = ty * BS + i;
            const double cr = colre[buf][r], ci = colim[buf][r];
            fr[i] = (r == p) ? 0.0 : ((r == pr) ? cpr : cr);
            fi[i] = (r == p) ? 0.0 : ((r == pr) ? cpi : ci);
        }
        #pragma unroll
        for (int i = 0; i < BS; i++)
            #pragma unroll
            for (int j = 0; j < BS; j++) {
                are[i][j] -= fr[i] * ur[j] - fi[i] * ui[j];
                aim[i][j] -= fr[i] * ui[j] + fi[i] * ur[j];
            }
        if (tx == pblk) {                               // column p: -f/pivot (pivot row fixed next)
            INV_SEL(poff, _Pragma("unroll") for (int i = 0; i < BS; i++) {
                are[i][I] = -(fr[i] * ir - fi[i] * ii); aim[i][I] = -(fr[i] * ii + fi[i] * ir); })
        }
        if (ty == pblk) {                               // pivot row: scaled row, 1/pivot in column p
            INV_SEL(poff, _Pragma("unroll") for (int j = 0; j < BS; j++) {
                const bool dg = (tx * BS + j == p);
                are[I][j] = dg ? ir : rpr[j]; aim[I][j] = dg ? ii : rpi[j]; })
        }
    }
#undef INV_SEL
    __syncthreads();
    if (t == 0) {   // compose the column swaps that undo the row interchanges
        for (int x = 0; x < NP; x++) idx[x] = x;
        for (int p = NP - 1; p >= 0; p--) { const int q = perm[p]; const int tmp = idx[p]; idx[p] = idx[q]; idx[q] = tmp; }
        for (int x = 0; x < NP; x++) outpos[idx[x]] = x;
    }
    __syncthreads();
    // stage the (column-permuted) inverse through LDS so that both output layouts are
    // written with coalesced stores; rows padded by one double against bank conflicts
    extern __shared__ double stage[];          // one plane at a time: NP x (NP+1)
    constexpr int LDP = NP + 1;
    double *A = LinvA + (size_t)n * 2 * pl, *T = LinvT + (size_t)n * 2 * pl;
    #pragma unroll
    for (int pass = 0; pass < 2; pass++) {
        #pragma unroll
        for (int j = 0; j < BS; j++) {
            const int oc = outpos[tx * BS + j];
            #pragma unroll
            for (int i = 0; i < BS; i++) stage[(ty * BS + i) * LDP + oc] = pass ? aim[i][j] : are[i][j];
        }
        __syncthreads();
        for (int e = t; e < NP * NP; e += NTH) {
            const int hi = e / NP, lo = e % NP;       // NP is a compile-time constant
            T[pass * pl + e] = stage[hi * LDP + lo];  // row-major: (r=hi, c=lo)
            A[pass * pl + e] = stage[lo * LDP + hi];  // column-major: (r=lo, c=hi)
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------
// K3: step propagator  P[n] = Linv[n+1] * R[n]   (n = 0..nt-2)
// (the implicit solve L(t_{n+1}) w_{n+1} = R(t_n) w_n of forward_evolution.jl:181-220,
//  done once for all right-hand sides).  Outputs P as panel (row-major; the
// left operand of the adjoint sweep as P^H) and as column-major planes (left
// operand of the forward sweep).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_propagator(const double *__restrict__ LinvA,
                                                    const double *__restrict__ R,
                                                    double *__restrict__ Pr, double *__restrict__ Pc,
                                                    int Np)
{
    __shared__ __attribute__((aligned(32))) double Bs[LV_KC][16 * LV_NG];
    const int n = blockIdx.y;
    const int ngroups = Np / 8;
    const int gtiles = (ngroups + LV_NG - 1) / LV_NG;
    const int gb = blockIdx.x % gtiles, rb4 = blockIdx.x / gtiles;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c16 = lane & 15, kk = lane >> 4;
    const int rb = rb4 * 4 + wave;
    const bool row_ok = rb * 16 < Np;
    const int arow = rb * 16 + c16;
    const int PW = 2 * Np;
    const size_t panel = (size_t)Np * PW, pl = (size_t)Np * Np;
    const double *Are = LinvA + (size_t)(n + 1) * 2 * pl, *Aim = Are + pl;
    const double *Bsrc = R + (size_t)n * panel;

    d4 acc[LV_NG];
    for (int g = 0; g < LV_NG; g++) acc[g] = (d4){0, 0, 0, 0};
    for (int kc = 0; kc < Np; kc += LV_KC) {
        {
            int r = threadIdx.x >> 4, c4 = (threadIdx.x & 15) * 4;
            int gcol = gb * 16 * LV_NG + c4;
            double4 v = make_double4(0, 0, 0, 0);
            if (gcol < PW) v = *reinterpret_cast<const double4 *>(Bsrc + (size_t)(kc + r) * PW + gcol);
            *reinterpret_cast<double4 *>(&Bs[r][c4]) = v;
        }
        double are[4], aim[4];
        if (row_ok) {
            #pragma unroll
            for (int s = 0; s < 4; s++) {
                size_t e = (size_t)arow + (size_t)Np * (kc + 4 * s + kk);
                are[s] = Are[e]; aim[s] = Aim[e];
            }
        }
        __syncthreads();
        if (row_ok) {
            #pragma unroll
            for (int s = 0; s < 4; s++)
                #pragma unroll
                for (int g = 0; g < LV_NG; g++) {
                    double b1, b2;
                    panel_b(&Bs[4 * s + kk][16 * g], c16, b1, b2);
                    acc[g] = MFMA(are[s], b1, acc[g]);
                    acc[g] = MFMA(aim[s], b2, acc[g]);
                }
        }
        __syncthreads();
    }
    if (!row_ok) return;
    double *Prn = Pr + (size_t)n * panel, *Pcn = Pc + (size_t)n * 2 * pl;
    #pragma unroll
    for (int g = 0; g < LV_NG; g++) {
        const int grp = gb * LV_NG + g;
        if (grp >= ngroups) continue;
        const int ccol = grp * 8 + (c16 & 7);
        const bool is_im = c16 >= 8;
        #pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = rb * 16 + kk + 4 * r;
            Prn[(size_t)row * PW + grp * 16 + c16] = acc[g][r];
            Pcn[(is_im ? pl : 0) + (size_t)row + (size_t)Np * ccol] = acc[g][r];
        }
    }
}

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
};

__device__ __forceinline__ const double *chain_matrix(const ChainArgs &a, int n)
{
    const size_t pl2 = (size_t)2 * a.Np * a.Np;
    if (a.pm_bpr) return a.Pmat + (size_t)(n / a.pm_bpr) * a.pm_chunk + (size_t)(n % a.pm_bpr) * pl2;
    return a.Pmat + (size_t)n * pl2;
}

template <int MODE>
__device__ __forceinline__ void chain_block_of(const ChainArgs &a, int &b, int &grp)
{
    if (MODE == 0) {
        // XCD-aware: the ngroups workgroups of one block share blockIdx%8, hence one XCD's L2,
        // because they all stream the same P_n (speed only; MI355X_MICROARCH.md "Workgroup dispatch").
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        b = xcd + 8 * (slot / a.ngroups);
        grp = slot % a.ngroups;
    } else {
        b = blockIdx.x / a.ngroups;
        grp = blockIdx.x % a.ngroups;
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

// NP > 0: compile-time size, 8 waves = (row block, K slice).  The state lives in LDS as the
// KSPLIT partial sums the waves produced (double buffered): the next step's B fragments are the
// sums of those partials, so a step needs ONE barrier.  The A fragments (and, for the adjoint, the
// forcing in accumulator layout) of the next PF steps are in flight in a register ring.
template <int NP, int MODE>
__global__ __launch_bounds__(512) void k_chain_fast(const ChainArgs a)
{
    constexpr bool ADJ = (MODE >= 2);
    constexpr int NRB = NP / 16;
    constexpr int KSPLIT = (8 / NRB) < (NP / 4) ? (8 / NRB) : (NP / 4);
    constexpr int NACT = NRB * KSPLIT;
    constexpr int KS = NP / 4 / KSPLIT;
    constexpr int PF = 3;
    constexpr int EPT = (NP * 16 + 511) / 512;           // panel elements per thread
    static_assert(NRB * 16 == NP && KS * KSPLIT * 4 == NP && NACT <= 8, "tile");
    __shared__ __attribute__((aligned(16))) double part[2][KSPLIT][NP * 16];

    int b, grp;
    chain_block_of<MODE>(a, b, grp);
    if (b >= a.nblocks) return;
    const int s0 = b * a.blen, e0 = (s0 + a.blen < a.S) ? s0 + a.blen : a.S;
    const int PWc = (MODE == 0) ? 2 * NP : 2 * a.cp;
    const size_t hstep = (size_t)NP * PWc;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int c16 = lane & 15, kk = lane >> 4;
    const bool active = wave < NACT;
    const int rb = wave / KSPLIT, kq = wave % KSPLIT;
    const int arow = rb * 16 + c16;
    const int nsteps = e0 - s0;

    // start state into part[0][0], zeros into the other partial slots
    #pragma unroll
    for (int t2 = 0; t2 < EPT; t2++) {
        const int e = tid + t2 * 512;
        if (e < NP * 16) {
            const int row = e >> 4, c = e & 15;
            double v;
            if (MODE == 0) v = (c < 8 && row == grp * 8 + c) ? 1.0 : 0.0;
            else if (MODE == 2) v = 0.0;
            else v = a.start[(size_t)b * a.start_stride + (size_t)row * PWc + grp * 16 + c];
            part[0][0][e] = v;
            #pragma unroll
            for (int w2 = 1; w2 < KSPLIT; w2++) part[0][w2][e] = 0.0;
        }
    }
    double rre[PF][KS], rim[PF][KS], rfo[PF][4];
    auto step_index = [&](int st) { return ADJ ? e0 - 1 - st : s0 + st; };
    auto issue = [&](int slot, int st) {                 // loads for step st into ring slot
        const int n = step_index(st);
        if (active) {
            const double *Pn = chain_matrix(a, n);
            #pragma unroll
            for (int i = 0; i < KS; i++) chain_a<ADJ>(Pn, NP, arow, (kq * KS + i) * 4 + kk, rre[slot][i], rim[slot][i]);
            if (ADJ && kq == 0) {                        // forcing f_n in accumulator layout
                const size_t fb = (size_t)(a.f_bpr ? n + n / a.f_bpr : n) * hstep;
                #pragma unroll
                for (int r = 0; r < 4; r++)
                    rfo[slot][r] = a.forcing[fb + (size_t)(rb * 16 + kk + 4 * r) * PWc + grp * 16 + c16];
            }
        }
    };
    #pragma unroll
    for (int q = 0; q < PF; q++) if (q < nsteps) issue(q, q);
    __syncthreads();

    int buf = 0;
    // one step on ring slot q; `refill` (>= 0) is the step whose operands replace the slot
    auto do_step = [&](int q, int st, int refill) {
        const int n = step_index(st);
        if (active) {
            d4 acc = (d4){0, 0, 0, 0};
            #pragma unroll
            for (int i = 0; i < KS; i++) {
                const int ko = ((kq * KS + i) * 4 + kk) * 16;
                double v1 = part[buf][0][ko + c16], v2 = part[buf][0][ko + (c16 ^ 8)];
                #pragma unroll
                for (int w2 = 1; w2 < KSPLIT; w2++) { v1 += part[buf][w2][ko + c16]; v2 += part[buf][w2][ko + (c16 ^ 8)]; }
                acc = MFMA(rre[q][i], v1, acc);
                acc = MFMA(rim[q][i], (c16 < 8) ? -v2 : v2, acc);
            }
            #pragma unroll
            for (int r = 0; r < 4; r++) {
                double v = acc[r];
                if (ADJ && kq == 0) v += rfo[q][r];
                part[buf ^ 1][kq][(rb * 16 + kk + 4 * r) * 16 + c16] = v;
            }
        }
        // history: part[buf] holds the state that the previous step produced (the start state is
        // not an output)
        if ((MODE == 1 || MODE == 3) && st > 0) {
            const int ncur = ADJ ? n + 1 : n;
            #pragma unroll
            for (int t2 = 0; t2 < EPT; t2++) {
                const int e = tid + t2 * 512;
                if (e < NP * 16) {
                    double v = part[buf][0][e];
                    #pragma unroll
                    for (int w2 = 1; w2 < KSPLIT; w2++) v += part[buf][w2][e];
                    a.out[(size_t)ncur * hstep + (size_t)(e >> 4) * PWc + grp * 16 + (e & 15)] = v;
                }
            }
        }
        if (refill >= 0) issue(q, refill);
        __syncthreads();
        buf ^= 1;
    };
    // steady state: whole groups of PF steps, no conditionals (keeps the compiler's vmcnt
    // bookkeeping exact so that PF-1 sets of loads really stay in flight); the refill index is
    // clamped, the last groups reload the final matrix harmlessly
    const int nfull = (nsteps / PF) * PF;
    for (int st0 = 0; st0 < nfull; st0 += PF) {
        #pragma unroll
        for (int q = 0; q < PF; q++) {
            const int nxt = st0 + q + PF;
            do_step(q, st0 + q, nxt < nsteps ? nxt : nsteps - 1);
        }
    }
    #pragma unroll
    for (int q = 0; q < PF; q++)
        if (nfull + q < nsteps) do_step(q, nfull + q, -1);
    // final state = sum of the partials in part[buf]
    const int nlast = ADJ ? s0 : e0;                     // its time index
    #pragma unroll
    for (int t2 = 0; t2 < EPT; t2++) {
        const int e = tid + t2 * 512;
        if (e < NP * 16) {
            double v = part[buf][0][e];
            #pragma unroll
            for (int w2 = 1; w2 < KSPLIT; w2++) v += part[buf][w2][e];
            const int row = e >> 4, c = e & 15;
            if ((MODE == 1 || MODE == 3) && nsteps > 0)
                a.out[(size_t)nlast * hstep + (size_t)row * PWc + grp * 16 + c] = v;
            if (MODE == 0) {
                const size_t pl = (size_t)NP * NP;
                double *pc = a.PiC + (size_t)b * 2 * pl, *pr = a.PiR + (size_t)b * 2 * pl;
                const int col = grp * 8 + (c & 7);
                pc[(c >= 8 ? pl : 0) + (size_t)row + (size_t)NP * col] = v;
                pr[(size_t)row * 2 * NP + grp * 16 + c] = v;
            }
            if (MODE == 2) a.phi[(size_t)b * hstep + (size_t)row * PWc + grp * 16 + c] = v;
        }
    }
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

template <int MODE>
static int launch_chain(const ChainArgs &a, hipStream_t stream)
{
    const int nwg = (MODE == 0) ? 8 * a.ngroups * ((a.nblocks + 7) / 8) : a.nblocks * a.ngroups;
    if (nwg <= 0) return 0;
    switch (a.Np) {
    case 16: hipLaunchKernelGGL((k_chain_fast<16, MODE>), dim3(nwg), dim3(512), 0, stream, a); break;
    case 32: hipLaunchKernelGGL((k_chain_fast<32, MODE>), dim3(nwg), dim3(512), 0, stream, a); break;
    case 64: hipLaunchKernelGGL((k_chain_fast<64, MODE>), dim3(nwg), dim3(512), 0, stream, a); break;
    default: {
        size_t shm = (size_t)2 * a.Np * 16 * sizeof(double);
        hipLaunchKernelGGL((k_chain_generic<MODE>), dim3(nwg), dim3(256), shm, stream, a);
    }
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
                                               int have_guard)
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
    if (threadIdx.x == 0 && (n > 0 || count_first)) {
        double tot = red[0] + red[1] + red[2] + red[3];
        atomicAdd(&scal[2], tot * trap * dt / tf);
    }
}

// K5 (fast path): W diagonal (the guard_projector of multi_qudit_systems.jl:316-349 is):
// elementwise forcing and penalty.  wd: the 2N diagonal entries.
__global__ __launch_bounds__(256) void k_guard_diag(const double *__restrict__ wd,
                                                    const double *__restrict__ hist,
                                                    double *__restrict__ forcing,
                                                    double *__restrict__ scal, int N, int Np, int cp,
                                                    int n_off, int nt_glob, int count_first, double dt, double tf)
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
    if (threadIdx.x == 0 && (n > 0 || count_first)) atomicAdd(&scal[2], (red[0] + red[1] + red[2] + red[3]) * trap * dt / tf);
}

// ---------------------------------------------------------------------------
// K6: overlaps and terminal right-hand side (infidelity.jl:13-17,
// eval_grad_discrete_adjoint.jl:22-40).  Single workgroup.
//   scal[0] = <w_N,R>, scal[1] = <w_N,T>;  y_N = (2/Ness^2)(a R + b T) + f_N
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_terminal(const double *__restrict__ hist,
                                                  const double *__restrict__ target,
                                                  const double *__restrict__ forcing,
                                                  double *__restrict__ yhist,
                                                  double *__restrict__ scal, int Np, int cp, int nt,
                                                  int n_ess, int have_target, int write_y,
                                                  double *__restrict__ y2, double *__restrict__ y3,
                                                  double *__restrict__ y4)
{
    __shared__ double red[8];
    const int PWc = 2 * cp;
    const size_t hstep = (size_t)Np * PWc;
    const double *w = hist + (size_t)(nt - 1) * hstep;
    double a = 0.0, b = 0.0;
    if (have_target) {
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
    for (int off = 32; off > 0; off >>= 1) { a += __shfl_down(a, off); b += __shfl_down(b, off); }
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = a; red[4 + (threadIdx.x >> 6)] = b; }
    __syncthreads();
    a = red[0] + red[1] + red[2] + red[3];
    b = red[4] + red[5] + red[6] + red[7];
    if (threadIdx.x == 0) { scal[0] = a; scal[1] = b; }
    if (!write_y) return;
    const double sc = 2.0 / ((double)n_ess * (double)n_ess);
    double *y = yhist + (size_t)(nt - 1) * hstep;
    const double *f = forcing + (size_t)(nt - 1) * hstep;
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

// ---------------------------------------------------------------------------
// K9: state derivatives at every time point (the stored history of
// forward_evolution.jl:172-179,:236-242):  psi_{j+1} = 1/(j+1) sum_{i<=j} A_{j-i} psi_i
// One workgroup per (time point, column group).  dpsi: [nt][m][Np][2cp].
// ---------------------------------------------------------------------------
template <int NOPS>
__global__ __launch_bounds__(256) void k_derivs(const double *__restrict__ ops,
                                                const double *__restrict__ tab,
                                                const double *__restrict__ hist,
                                                double *__restrict__ dpsi, int Np, int cp,
                                                int n_ops, int m, double *__restrict__ gpanels)
{
    extern __shared__ double lds_panels[];              // (m+1) panels [Np][16] ...
    const int n = blockIdx.y, grp = blockIdx.x;
    // ... or, when they do not fit in LDS (large N), a slab of global scratch per workgroup
    double *smem = gpanels ? gpanels + ((size_t)n * gridDim.x + grp) * (size_t)(m + 1) * Np * 16 : lds_panels;
    const int PWc = 2 * cp;
    const size_t hstep = (size_t)Np * PWc;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    const int c16 = lane & 15, kk = lane >> 4;
    const size_t ps = (size_t)Np * 16;
    for (int e = threadIdx.x; e < Np * 16; e += blockDim.x)
        smem[e] = hist[(size_t)n * hstep + (size_t)(e >> 4) * PWc + grp * 16 + (e & 15)];
    __threadfence_block();
    __syncthreads();
    for (int j = 0; j < m; j++) {
        for (int rb = wave; rb * 16 < Np; rb += nw) {
            d4 acc = (d4){0, 0, 0, 0};
            const int arow = rb * 16 + c16;
            for (int i = 0; i <= j; i++) {
                OpCoef cf;
                load_coef(cf, tab, n, j - i, m, n_ops);
                const double *src = smem + (size_t)i * ps;
                for (int k0 = 0; k0 < Np; k0 += 4) {
                    double are, aim, b1, b2;
                    assembled_a<NOPS>(ops, Np, n_ops, cf, arow, k0 + kk, are, aim);
                    panel_b(src + (size_t)(k0 + kk) * 16, c16, b1, b2);
                    acc = MFMA(are, b1, acc);
                    acc = MFMA(aim, b2, acc);
                }
            }
            const double inv = 1.0 / (double)(j + 1);
            #pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = rb * 16 + kk + 4 * r;
                const double v = acc[r] * inv;
                smem[(size_t)(j + 1) * ps + (size_t)row * 16 + c16] = v;
                dpsi[(((size_t)n * m + j) * Np + row) * PWc + grp * 16 + c16] = v;
            }
        }
        __threadfence_block();
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------
// K10: gradient scalars.  Per (time point, column group):
//   seeds   g_j = c_j dt^j lambda_{n+1} [n<=nt-2]  -  c_j (-dt)^j lambda_n [n>=1]
//   sweep   j = m..2, i = 1..j-1:  g_i += (1/j) A_{j-1-i}^H g_j     (A^H = -A)
//   sigma   sigP[k][d] += (1/j) <(dA/dp_k) psi_i, g_j>,  sigQ likewise, d = j-1-i
// This is the O(m^2) reverse form of accumulate_gradient_arbitrary_fast! /
// recursive_magic! (eval_grad_discrete_adjoint.jl:582-726); inner products as
// compute_inner_prod_S!/K! (:764-800).  sigma: [nt][n_ops][m][2] (atomicAdd
// across column groups).
// ---------------------------------------------------------------------------
template <int NOPS>
__global__ __launch_bounds__(256) void k_gradsweep(const double *__restrict__ ops,
                                                   const double *__restrict__ tab,
                                                   const double *__restrict__ hist,
                                                   const double *__restrict__ dpsi,
                                                   const double *__restrict__ lam,
                                                   double *__restrict__ sigma,
                                                   const double *__restrict__ cw, int Np, int cp,
                                                   int n_ops, int m, int nt, double *__restrict__ gpanels)
{
    extern __shared__ double smem[];     // psi_0..psi_{m-1} (m panels), g_1..g_m (m panels), sig[n_ops*m*2]
    const int n = blockIdx.y, grp = blockIdx.x;
    const int PWc = 2 * cp;
    const size_t hstep = (size_t)Np * PWc;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    const int c16 = lane & 15, kk = lane >> 4;
    const size_t ps = (size_t)Np * 16;
    // panels in LDS, or (large N) in a slab of global scratch per workgroup; sig always in LDS
    double *pbase = gpanels ? gpanels + ((size_t)n * gridDim.x + grp) * (size_t)(2 * m) * Np * 16 : smem;
    double *psi = pbase, *gs = pbase + (size_t)m * ps;   // gs[(j-1)*ps]
    double *sig = gpanels ? smem : gs + (size_t)m * ps;
    const size_t pl = (size_t)Np * Np;

    for (int e = threadIdx.x; e < Np * 16; e += blockDim.x) {
        const size_t src = (size_t)(e >> 4) * PWc + grp * 16 + (e & 15);
        psi[e] = hist[(size_t)n * hstep + src];
        for (int i = 1; i < m; i++) psi[(size_t)i * ps + e] = dpsi[(((size_t)n * m + (i - 1)) * Np) * PWc + src];
        const double ln = (n >= 1) ? lam[(size_t)n * hstep + src] : 0.0;
        const double lx = (n <= nt - 2) ? lam[(size_t)(n + 1) * hstep + src] : 0.0;
        for (int j = 1; j <= m; j++) gs[(size_t)(j - 1) * ps + e] = cw[2 * j] * lx - cw[2 * j + 1] * ln;
    }
    for (int e = threadIdx.x; e < n_ops * m * 2; e += blockDim.x) sig[e] = 0.0;
    __threadfence_block();
    __syncthreads();

    // reverse sweep
    for (int j = m; j >= 2; j--) {
        const double *src = gs + (size_t)(j - 1) * ps;
        for (int rb = wave; rb * 16 < Np; rb += nw) {
            const int arow = rb * 16 + c16;
            for (int i = 1; i <= j - 1; i++) {
                OpCoef cf;
                load_coef(cf, tab, n, j - 1 - i, m, n_ops);
                d4 acc = (d4){0, 0, 0, 0};
                for (int k0 = 0; k0 < Np; k0 += 4) {
                    double are, aim, b1, b2;
                    assembled_a<NOPS>(ops, Np, n_ops, cf, arow, k0 + kk, are, aim);
                    panel_b(src + (size_t)(k0 + kk) * 16, c16, b1, b2);
                    acc = MFMA(are, b1, acc);
                    acc = MFMA(aim, b2, acc);
                }
                const double sc = -1.0 / (double)j;     // A^H = -A
                #pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int row = rb * 16 + kk + 4 * r;
                    gs[(size_t)(i - 1) * ps + (size_t)row * 16 + c16] += sc * acc[r];
                }
            }
        }
        __threadfence_block();
        __syncthreads();
    }

    // inner products
    for (int o = 0; o < n_ops; o++) {
        const double *Asym = ops + (size_t)(2 + 2 * o) * pl, *Sym = ops + (size_t)(3 + 2 * o) * pl;
        for (int i = 0; i < m; i++) {
            const double *src = psi + (size_t)i * ps;
            for (int rb = wave; rb * 16 < Np; rb += nw) {
                const int arow = rb * 16 + c16;
                d4 U = (d4){0, 0, 0, 0}, V = (d4){0, 0, 0, 0};
                for (int k0 = 0; k0 < Np; k0 += 4) {
                    const size_t e = (size_t)arow + (size_t)Np * (k0 + kk);
                    const double b1 = src[(size_t)(k0 + kk) * 16 + c16];
                    U = MFMA(Sym[e], b1, U);
                    V = MFMA(Asym[e], b1, V);
                }
                for (int j = i + 1; j <= m; j++) {
                    const double *gj = gs + (size_t)(j - 1) * ps;
                    double sp = 0.0, sq = 0.0;
                    #pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int row = rb * 16 + kk + 4 * r;
                        double g1, g2;
                        panel_b(gj + (size_t)row * 16, c16, g1, g2);
                        sq += V[r] * g1;              // Re<V, g>
                        sp += U[r] * g2;              // Re<-iU, g> = Uim*gre - Ure*gim
                    }
                    for (int off = 32; off > 0; off >>= 1) { sp += __shfl_down(sp, off); sq += __shfl_down(sq, off); }
                    if (lane == 0) {
                        const int d = j - 1 - i;
                        atomicAdd(&sig[(o * m + d) * 2], sp / (double)j);
                        atomicAdd(&sig[(o * m + d) * 2 + 1], sq / (double)j);
                    }
                }
            }
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < n_ops * m * 2; e += blockDim.x)
        atomicAdd(&sigma[(size_t)n * n_ops * m * 2 + e], sig[e]);
}

// ---------------------------------------------------------------------------
// K9+K10 fused (fast path, Np = 64): everything the gradient needs at one time point, one
// launch, no derivative history in HBM.  Workgroup = (time point, group of 8 columns), 4 waves =
// 4 row blocks.  psi_0..psi_{m-1} live in LDS; g_1..g_m live in REGISTERS in accumulator layout
// (each wave owns its 16 rows) and only the g_j currently acting as right operand is staged in
// LDS.  All passes share one register ring of operator elements that wraps around across passes
// (the (row,k) elements are the same in every pass), so the ring never drains.
//   D passes  psi_{j+1} = 1/(j+1) sum_{i<=j} A_{j-i} psi_i          j = 0..m-2
//   G passes  g_i += (1/j) A_{j-1-i}^H g_j (A^H = -A), i = 1..j-1   j = m..2
//   S passes  per operator o: U_i = Sym_o psi_i, V_i = Asym_o psi_i, then the inner products
// ---------------------------------------------------------------------------
template <int M, int NOPS>
__global__ __launch_bounds__(256) void k_gradpoint64(const double *__restrict__ ops,
                                                     const double *__restrict__ tab,
                                                     const double *__restrict__ hist,
                                                     const double *__restrict__ lam,
                                                     double *__restrict__ sigma,
                                                     const double *__restrict__ cw, int cp, int nt,
                                                     int n_ops)
{
    constexpr int NP = 64, NKS = NP / 4, PS = NP * 16, RD = 4;
    extern __shared__ double smem[];
    double *psi = smem;                         // [M][PS]
    double *gsrc = smem + (size_t)M * PS;       // [PS]
    double *sig = gsrc + PS;                    // [n_ops*M*2]
    const int n = blockIdx.y, grp = blockIdx.x;
    const int PWc = 2 * cp;
    const size_t hstep = (size_t)NP * PWc, pl = (size_t)NP * NP;
    const int tid = threadIdx.x, rb = tid >> 6, lane = tid & 63;
    const int c16 = lane & 15, kk = lane >> 4;
    const int arow = rb * 16 + c16;

    for (int e = tid; e < PS; e += 256)
        psi[e] = hist[(size_t)n * hstep + (size_t)(e >> 4) * PWc + grp * 16 + (e & 15)];
    for (int e = tid; e < n_ops * M * 2; e += 256) sig[e] = 0.0;

    // seeds in accumulator layout
    d4 g[M];
    {
        double ln[4], lx[4];
        #pragma unroll
        for (int r = 0; r < 4; r++) {
            const size_t o = (size_t)(rb * 16 + kk + 4 * r) * PWc + grp * 16 + c16;
            ln[r] = (n >= 1) ? lam[(size_t)n * hstep + o] : 0.0;
            lx[r] = (n <= nt - 2) ? lam[(size_t)(n + 1) * hstep + o] : 0.0;
        }
        #pragma unroll
        for (int j = 1; j <= M; j++)
            #pragma unroll
            for (int r = 0; r < 4; r++) g[j - 1][r] = cw[2 * j] * lx[r] - cw[2 * j + 1] * ln[r];
    }
    // coefficients of derivative orders 0..M-2 (uniform -> scalar registers)
    constexpr int ND = (M > 1) ? M - 1 : 1;
    double cfr[ND][CF_STRIDE];
    #pragma unroll
    for (int d = 0; d < ND; d++) {
        cfr[d][0] = (d == 0) ? 1.0 : 0.0;
        #pragma unroll
        for (int o = 0; o < NOPS_LIM(NOPS); o++) {
            const bool on = NOPS_ON(NOPS, o, n_ops);
            cfr[d][1 + 2 * o] = on ? tab[(((size_t)n * (M + 1) + d) * n_ops + o) * 2] : 0.0;
            cfr[d][2 + 2 * o] = on ? tab[(((size_t)n * (M + 1) + d) * n_ops + o) * 2 + 1] : 0.0;
        }
    }
    // operator ring, wraps around over k-steps and passes
    OpVals ring[RD];
    auto ring_load = [&](int slot, int ks) {
        load_opvals<NOPS>(ring[slot], ops, NP, n_ops, (size_t)arow + (size_t)NP * ((ks & (NKS - 1)) * 4 + kk));
    };
    if (M > 1) {
        #pragma unroll
        for (int q = 0; q < RD - 1; q++) ring_load(q, q);
    }
    __syncthreads();

    // ---- D passes
    #pragma unroll
    for (int j = 0; j + 1 < M; j++) {
        d4 acc = (d4){0, 0, 0, 0};
        for (int ks4 = 0; ks4 < NKS; ks4 += RD) {
            #pragma unroll
            for (int q = 0; q < RD; q++) {
                const int ks = ks4 + q, k = ks * 4 + kk;
                ring_load((q + RD - 1) % RD, ks + RD - 1);
                #pragma unroll
                for (int i = 0; i <= j; i++) {
                    double are, aim, b1, b2;
                    combine_opvals<NOPS>(ring[q], cfr[j - i], n_ops, are, aim);
                    panel_b(psi + (size_t)i * PS + (size_t)k * 16, c16, b1, b2);
                    acc = MFMA(are, b1, acc);
                    acc = MFMA(aim, b2, acc);
                }
            }
        }
        const double inv = 1.0 / (double)(j + 1);
        #pragma unroll
        for (int r = 0; r < 4; r++) psi[(size_t)(j + 1) * PS + (size_t)(rb * 16 + kk + 4 * r) * 16 + c16] = acc[r] * inv;
        __syncthreads();
    }

    // ---- G passes
    #pragma unroll
    for (int j = M; j >= 2; j--) {
        #pragma unroll
        for (int r = 0; r < 4; r++) gsrc[(size_t)(rb * 16 + kk + 4 * r) * 16 + c16] = g[j - 1][r];
        __syncthreads();
        d4 t[M];
        #pragma unroll
        for (int i = 0; i < M; i++) t[i] = (d4){0, 0, 0, 0};
        for (int ks4 = 0; ks4 < NKS; ks4 += RD) {
            #pragma unroll
            for (int q = 0; q < RD; q++) {
                const int ks = ks4 + q, k = ks * 4 + kk;
                ring_load((q + RD - 1) % RD, ks + RD - 1);
                double b1, b2;
                panel_b(gsrc + (size_t)k * 16, c16, b1, b2);
                #pragma unroll
                for (int i = 1; i <= j - 1; i++) {
                    double are, aim;
                    combine_opvals<NOPS>(ring[q], cfr[j - 1 - i], n_ops, are, aim);
                    t[i] = MFMA(are, b1, t[i]);
                    t[i] = MFMA(aim, b2, t[i]);
                }
            }
        }
        const double sc = -1.0 / (double)j;              // A^H = -A
        #pragma unroll
        for (int i = 1; i <= j - 1; i++)
            #pragma unroll
            for (int r = 0; r < 4; r++) g[i - 1][r] += sc * t[i][r];
        __syncthreads();                                 // gsrc is rewritten by the next pass
    }

    // [-g_im | g_re] partner of every g_j, for the <(dA/dp) psi, g> products
    d4 gs[M];
    #pragma unroll
    for (int j = 0; j < M; j++)
        #pragma unroll
        for (int r = 0; r < 4; r++) {
            const double o = __shfl_xor(g[j][r], 8);
            gs[j][r] = (c16 < 8) ? -o : o;
        }

    // ---- S passes
    #pragma unroll
    for (int o = 0; o < NOPS_LIM(NOPS); o++) {
        if (!NOPS_ON(NOPS, o, n_ops)) continue;
        const double *Asym = ops + (size_t)(2 + 2 * o) * pl, *Sym = ops + (size_t)(3 + 2 * o) * pl;
        d4 U[M], V[M];
        #pragma unroll
        for (int i = 0; i < M; i++) { U[i] = (d4){0, 0, 0, 0}; V[i] = (d4){0, 0, 0, 0}; }
        double rs[RD], ra[RD];
        #pragma unroll
        for (int q = 0; q < RD - 1; q++) {
            const size_t e = (size_t)arow + (size_t)NP * (q * 4 + kk);
            rs[q] = Sym[e]; ra[q] = Asym[e];
        }
        for (int ks4 = 0; ks4 < NKS; ks4 += RD) {
            #pragma unroll
            for (int q = 0; q < RD; q++) {
                const int ks = ks4 + q, k = ks * 4 + kk;
                {
                    const size_t e = (size_t)arow + (size_t)NP * ((((ks + RD - 1) & (NKS - 1)) * 4) + kk);
                    rs[(q + RD - 1) % RD] = Sym[e]; ra[(q + RD - 1) % RD] = Asym[e];
                }
                #pragma unroll
                for (int i = 0; i < M; i++) {
                    const double b1 = psi[(size_t)i * PS + (size_t)k * 16 + c16];
                    U[i] = MFMA(rs[q], b1, U[i]);
                    V[i] = MFMA(ra[q], b1, V[i]);
                }
            }
        }
        double sp[M], sq[M];
        #pragma unroll
        for (int d = 0; d < M; d++) { sp[d] = 0.0; sq[d] = 0.0; }
        #pragma unroll
        for (int i = 0; i < M; i++)
            #pragma unroll
            for (int j = i + 1; j <= M; j++) {
                double ap = 0.0, aq = 0.0;
                #pragma unroll
                for (int r = 0; r < 4; r++) { ap += U[i][r] * gs[j - 1][r]; aq += V[i][r] * g[j - 1][r]; }
                sp[j - 1 - i] += ap / (double)j;
                sq[j - 1 - i] += aq / (double)j;
            }
        #pragma unroll
        for (int d = 0; d < M; d++) {
            #pragma unroll
            for (int off = 32; off > 0; off >>= 1) { sp[d] += __shfl_down(sp[d], off); sq[d] += __shfl_down(sq[d], off); }
            if (lane == 0) { atomicAdd(&sig[(o * M + d) * 2], sp[d]); atomicAdd(&sig[(o * M + d) * 2 + 1], sq[d]); }
        }
    }
    __syncthreads();
    for (int e = tid; e < n_ops * M * 2; e += 256)
        atomicAdd(&sigma[(size_t)n * n_ops * M * 2 + e], sig[e]);
}

template <int M, int NOPS>
static int launch_gradpoint64(const qgdk_ctx *c)
{
    const size_t shm = ((size_t)(M + 1) * 64 * 16 + (size_t)c->n_ops * M * 2) * sizeof(double);
    SET_LDS_ONCE((k_gradpoint64<M, NOPS>), shm);
    hipLaunchKernelGGL((k_gradpoint64<M, NOPS>), dim3(c->cp / 8, c->nt), dim3(256), shm, c->stream, c->ops, c->tab,
                       c->hist, c->lam, c->sigma, c->cw, c->cp, c->nt, c->n_ops);
    return (int)hipGetLastError();
}

template <int M>
static int launch_gradpoint64_m(const qgdk_ctx *c)
{
#define CALL_GP(N) return launch_gradpoint64<M, N>(c)
    DISPATCH_NOPS(c->n_ops, CALL_GP)
#undef CALL_GP
    return 0;
}

// ---------------------------------------------------------------------------
// K11: gradient contraction  grad[off_k + l] = - sum_{n,d} Gp[k][n][d][l] sigP[n][k][d] + Gq.. sigQ..
// (the "grad_slice .-= contrib" of eval_grad_discrete_adjoint.jl:642-643)
// grid: (ceil(nc_max/64), n_ops, NSPLIT); atomicAdd over the time splits.
// ---------------------------------------------------------------------------
#define CT_CHUNK 16
__global__ __launch_bounds__(256) void k_contract(const double *__restrict__ G, const int64_t *__restrict__ goff,
                                                  const int32_t *__restrict__ ncoef,
                                                  const int32_t *__restrict__ poff,
                                                  const double *__restrict__ sigma,
                                                  double *__restrict__ grad, int nt, int m, int n_ops)
{
    // grid (time chunks, n_ops, coefficient tiles of 64); thread = (coefficient, time sub-slot)
    __shared__ double red[4][64];
    const int k = blockIdx.y;
    const int nc = ncoef[k];
    const int l = blockIdx.z * 64 + (threadIdx.x & 63), sub = threadIdx.x >> 6;
    const double *gp = G + goff[k];
    const double *gq = gp + (size_t)nt * (m + 1) * nc;
    double s = 0.0;
    if (l < nc) {
        const int n1 = min(nt, (int)(blockIdx.x + 1) * CT_CHUNK);
        for (int n = blockIdx.x * CT_CHUNK + sub; n < n1; n += 4)
            for (int d = 0; d < m; d++) {
                const double sp = sigma[(((size_t)n * n_ops + k) * m + d) * 2];
                const double sq = sigma[(((size_t)n * n_ops + k) * m + d) * 2 + 1];
                s += gp[((size_t)n * (m + 1) + d) * nc + l] * sp + gq[((size_t)n * (m + 1) + d) * nc + l] * sq;
            }
    }
    red[sub][threadIdx.x & 63] = s;
    __syncthreads();
    if (sub == 0 && l < nc)
        atomicAdd(&grad[poff[k] + l], -(red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]));
}

// ---------------------------------------------------------------------------
// Test hook: out = (+/-) A_d(t_n) * in for a panel of columns (apply_hamiltonian!,
// hermite.jl:556-588, batched over all initial-condition columns).
// ---------------------------------------------------------------------------
template <int NOPS>
__global__ __launch_bounds__(256) void k_apply(const double *__restrict__ ops,
                                               const double *__restrict__ tab,
                                               const double *__restrict__ in,
                                               double *__restrict__ out, int Np, int cp, int n_ops,
                                               int m, int n, int d, double sign)
{
    extern __shared__ double smem[];
    const int grp = blockIdx.x;
    const int PWc = 2 * cp;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    const int c16 = lane & 15, kk = lane >> 4;
    for (int e = threadIdx.x; e < Np * 16; e += blockDim.x)
        smem[e] = in[(size_t)(e >> 4) * PWc + grp * 16 + (e & 15)];
    __syncthreads();
    OpCoef cf;
    load_coef(cf, tab, n, d, m, n_ops);
    for (int rb = wave; rb * 16 < Np; rb += nw) {
        d4 acc = (d4){0, 0, 0, 0};
        const int arow = rb * 16 + c16;
        for (int k0 = 0; k0 < Np; k0 += 4) {
            double are, aim, b1, b2;
            assembled_a<NOPS>(ops, Np, n_ops, cf, arow, k0 + kk, are, aim);
            panel_b(smem + (size_t)(k0 + kk) * 16, c16, b1, b2);
            acc = MFMA(are, b1, acc);
            acc = MFMA(aim, b2, acc);
        }
        #pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = rb * 16 + kk + 4 * r;
            out[(size_t)row * PWc + grp * 16 + c16] = sign * acc[r];
        }
    }
}

// ---------------------------------------------------------------------------
// launchers (called from qgd_api.cpp)
// ---------------------------------------------------------------------------

extern "C" {

int qgdk_tables(const qgdk_ctx *c, const double *pcof)
{
    int total = c->nt * (c->m + 1) * c->n_ops * 2;
    hipLaunchKernelGGL(k_tables, dim3((total + 255) / 256 + 1), dim3(256), 0, c->stream, c->G, c->goff, c->ncoef,
                       c->poff, pcof, c->tab, c->nt, c->m, c->n_ops, c->scal, c->status);
    return (int)hipGetLastError();
}

int qgdk_tables_from_host(const qgdk_ctx *c, const double *pt, const double *qt)
{
    int total = c->nt * (c->m + 1) * c->n_ops * 2;
    hipLaunchKernelGGL(k_tables_from_host, dim3((total + 255) / 256), dim3(256), 0, c->stream, pt, qt, c->tab,
                       c->nt, c->m, c->n_ops);
    return (int)hipGetLastError();
}

int qgdk_build_LR(const qgdk_ctx *c)
{
    if (c->Np == 64) {           // fused LDS-resident path (order <= 10: D_1..D_{m-1} slabs fit in LDS)
        switch (c->m) {
        case 1: return launch_build_LR64<1>(c);
        case 2: return launch_build_LR64<2>(c);
        case 3: return launch_build_LR64<3>(c);
        case 4: return launch_build_LR64<4>(c);
        case 5: return launch_build_LR64<5>(c);
        default: break;
        }
    }
    const int ngroups = c->Np / 8;
    const int gtiles = (ngroups + LV_NG - 1) / LV_NG;
    const int rtiles = (c->Np + 63) / 64;
    for (int j = 0; j < c->m; j++) {
        double cL = c->cw_host[2 * (j + 1) + 1], cR = c->cw_host[2 * (j + 1)];
#define CALL_LV(N) hipLaunchKernelGGL((k_level<N>), dim3(gtiles * rtiles, c->nt), dim3(256), 0, c->stream, c->ops, c->tab, \
                                      c->D, c->L, c->R, c->Np, c->n_ops, c->m, j, cL, cR)
        DISPATCH_NOPS(c->n_ops, CALL_LV)
#undef CALL_LV
    }
    return (int)hipGetLastError();
}

int qgdk_inverse(const qgdk_ctx *c)
{
    const int nmat = c->nt - 1;
    switch (c->Np) {
    case 16: SET_LDS_ONCE((k_inverse_reg<16, 16>), 2176); hipLaunchKernelGGL((k_inverse_reg<16, 16>), dim3(nmat), dim3(256), 2176, c->stream, c->L, c->LinvA, c->LinvT, 1, c->status); return (int)hipGetLastError();
    case 32: SET_LDS_ONCE((k_inverse_reg<32, 16>), 8448); hipLaunchKernelGGL((k_inverse_reg<32, 16>), dim3(nmat), dim3(256), 8448, c->stream, c->L, c->LinvA, c->LinvT, 1, c->status); return (int)hipGetLastError();
    case 48: SET_LDS_ONCE((k_inverse_reg<48, 16>), 18816); hipLaunchKernelGGL((k_inverse_reg<48, 16>), dim3(nmat), dim3(256), 18816, c->stream, c->L, c->LinvA, c->LinvT, 1, c->status); return (int)hipGetLastError();
    case 64:   // (a one-wave variant, <64, 8>, measured 0.35 ms vs 0.22 ms: latency-bound at one wave per SIMD)
        SET_LDS_ONCE((k_inverse_reg<64, 16>), 33280); hipLaunchKernelGGL((k_inverse_reg<64, 16>), dim3(nmat), dim3(256), 33280, c->stream, c->L, c->LinvA, c->LinvT, 1, c->status); return (int)hipGetLastError();
    default: break;
    }
    const size_t pl = (size_t)c->Np * c->Np;
    size_t aux = (size_t)(3 * c->Np + 16) * sizeof(double);
    size_t mat = 2 * pl * sizeof(double);
    int use_lds = (aux + mat <= 150 * 1024) ? 1 : 0;
    size_t shm = aux + (use_lds ? mat : 0);
    if (use_lds) {
        HIPCHK(hipFuncSetAttribute((const void *)k_inverse, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
        hipLaunchKernelGGL(k_inverse, dim3(c->nt - 1), dim3(256), shm, c->stream, c->L, c->LinvA, c->LinvT,
                           (double *)nullptr, c->Np, 1, 1, c->status);
    } else {
        // global scratch: process in batches of inv_batch matrices
        for (int n0 = 1; n0 < c->nt; n0 += c->inv_batch) {
            int nb = (c->nt - n0 < c->inv_batch) ? c->nt - n0 : c->inv_batch;
            hipLaunchKernelGGL(k_inverse, dim3(nb), dim3(256), shm, c->stream, c->L, c->LinvA, c->LinvT,
                               c->inv_scratch, c->Np, n0, 0, c->status);
        }
    }
    return (int)hipGetLastError();
}

int qgdk_propagator(const qgdk_ctx *c)
{
    const int ngroups = c->Np / 8;
    const int gtiles = (ngroups + LV_NG - 1) / LV_NG;
    const int rtiles = (c->Np + 63) / 64;
    hipLaunchKernelGGL(k_propagator, dim3(gtiles * rtiles, c->nt - 1), dim3(256), 0, c->stream, c->LinvA, c->R,
                       c->Pr, c->Pc, c->Np);
    return (int)hipGetLastError();
}

// Exchange layout (DESIGN.md "Multi-GPU"): PiX = one chunk per rank, chunk = [bpr x PiC | bpr x PiR];
// phiX = one chunk per rank, chunk = [bpr x phi | 1 x y_N (last rank only)].
static inline size_t pix_chunk(const qgdk_ctx *c) { return (size_t)2 * c->bpr * 2 * c->Np * c->Np; }
static inline size_t phix_chunk(const qgdk_ctx *c) { return (size_t)(c->bpr + 1) * c->Np * 2 * c->cp; }

// forward, phase (i): block propagators of the owned blocks into this rank's chunk of PiX
int qgdk_forward_blocks(const qgdk_ctx *c)
{
    ChainArgs a{};
    a.Np = c->Np; a.cp = c->cp; a.S = c->nt - 1; a.Pmat = c->Pc;
    a.PiC = c->PiX + (size_t)c->part_rank * pix_chunk(c);
    a.PiR = a.PiC + (size_t)c->bpr * 2 * c->Np * c->Np;
    a.nblocks = c->blk_hi - c->blk_lo; a.blen = c->scan_blen; a.ngroups = c->Np / 8;
    return launch_chain<0>(a, c->stream);
}

// forward, phases (ii)+(iii): boundary states over ALL blocks (every rank), then the owned blocks.
// Phase (ii) is itself a scan over the B block propagators when B is large (second level:
// super-blocks of scan_g blocks): chain length g + B2 + g instead of B.
int qgdk_forward_finish(const qgdk_ctx *c)
{
    const size_t hstep = (size_t)c->Np * 2 * c->cp;
    const int B = c->scan_blocks, B2 = c->scan_blocks2, g = c->scan_g;
    int rc;
    // bnd[0] = bnd2[0] = psi_0 were written when the grid was allocated (the initial state is constant)
    if (B2 <= 1) {
        ChainArgs s2{};
        s2.Np = c->Np; s2.cp = c->cp; s2.S = B; s2.Pmat = c->PiX; s2.pm_bpr = c->bpr; s2.pm_chunk = (long long)pix_chunk(c);
        s2.start = c->bnd; s2.start_stride = 0; s2.out = c->bnd; s2.nblocks = 1; s2.blen = B; s2.ngroups = c->cp / 8;
        if ((rc = launch_chain<1>(s2, c->stream))) return rc;
    } else {
        ChainArgs a2{};   // (ii-a) super-block propagators from the block propagators
        a2.Np = c->Np; a2.cp = c->cp; a2.S = B; a2.Pmat = c->PiX; a2.pm_bpr = c->bpr; a2.pm_chunk = (long long)pix_chunk(c);
        a2.PiC = c->PiC2; a2.PiR = c->PiR2; a2.nblocks = B2; a2.blen = g; a2.ngroups = c->Np / 8;
        if ((rc = launch_chain<0>(a2, c->stream))) return rc;
        ChainArgs b2{};   // (ii-b) states at super-block starts
        b2.Np = c->Np; b2.cp = c->cp; b2.S = B2; b2.Pmat = c->PiC2; b2.start = c->bnd2; b2.start_stride = 0; b2.out = c->bnd2;
        b2.nblocks = 1; b2.blen = B2; b2.ngroups = c->cp / 8;
        if ((rc = launch_chain<1>(b2, c->stream))) return rc;
        ChainArgs c2{};   // (ii-c) states at every block start
        c2.Np = c->Np; c2.cp = c->cp; c2.S = B; c2.Pmat = c->PiX; c2.pm_bpr = c->bpr; c2.pm_chunk = (long long)pix_chunk(c);
        c2.start = c->bnd2; c2.start_stride = (long long)hstep; c2.out = c->bnd; c2.nblocks = B2; c2.blen = g; c2.ngroups = c->cp / 8;
        if ((rc = launch_chain<1>(c2, c->stream))) return rc;
    }
    if (c->blk_lo > 0)   // rank 0's hist[0] = psi_0 is constant
        HIPCHK(hipMemcpyAsync(c->hist, c->bnd + (size_t)c->blk_lo * hstep, hstep * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    ChainArgs s3{};
    s3.Np = c->Np; s3.cp = c->cp; s3.S = c->nt - 1; s3.Pmat = c->Pc; s3.start = c->bnd + (size_t)c->blk_lo * hstep;
    s3.start_stride = (long long)hstep; s3.out = c->hist; s3.nblocks = c->blk_hi - c->blk_lo; s3.blen = c->scan_blen;
    s3.ngroups = c->cp / 8;
    return launch_chain<1>(s3, c->stream);
}

int qgdk_guard(const qgdk_ctx *c)
{
    const int count_first = (c->n_off == 0) ? 1 : 0;
    if (c->have_guard == 2) {   // diagonal projector
        hipLaunchKernelGGL(k_guard_diag, dim3(c->nt), dim3(256), 0, c->stream, c->guard_diag, c->hist, c->forcing,
                           c->scal, c->N, c->Np, c->cp, c->n_off, c->nt_glob, count_first, c->dt, c->tf);
        return (int)hipGetLastError();
    }
    hipLaunchKernelGGL(k_guard, dim3(c->nt), dim3(256), 0, c->stream, c->guard, c->hist, c->forcing, c->scal,
                       c->N, c->Np, c->c, c->cp, c->n_off, c->nt_glob, count_first, c->dt, c->tf, c->have_guard);
    return (int)hipGetLastError();
}

int qgdk_terminal(const qgdk_ctx *c, int write_y)
{
    const size_t hstep = (size_t)c->Np * 2 * c->cp;
    double *slot = c->phiX + (size_t)c->part_rank * phix_chunk(c) + (size_t)c->bpr * hstep;
    hipLaunchKernelGGL(k_terminal, dim3(1), dim3(256), 0, c->stream, c->hist, c->target, c->forcing, c->yhist,
                       c->scal, c->Np, c->cp, c->nt, c->n_ess, c->have_target, write_y, slot,
                       c->bndY + (size_t)c->scan_blocks * hstep, c->bndY2 + (size_t)c->scan_blocks2 * hstep);
    return (int)hipGetLastError();
}

// adjoint, phase (i): affine parts phi_b of the owned blocks into this rank's chunk of phiX; the
// rank that owns the final time puts y_N (written by k_terminal into yhist) in its extra slot
int qgdk_adjoint_blocks(const qgdk_ctx *c)
{
    const size_t hstep = (size_t)c->Np * 2 * c->cp;
    double *own = c->phiX + (size_t)c->part_rank * phix_chunk(c);
    ChainArgs a{};
    a.Np = c->Np; a.cp = c->cp; a.S = c->nt - 1; a.Pmat = c->Pr; a.forcing = c->forcing; a.phi = own;
    a.nblocks = c->blk_hi - c->blk_lo; a.blen = c->scan_blen; a.ngroups = c->cp / 8;
    int rc = launch_chain<2>(a, c->stream);
    if (rc) return rc;
    // the last rank's k_terminal wrote y_N into its extra slot (and into bndY/bndY2/yhist) directly;
    // the other ranks' extra slots are never read
    return 0;
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
    double *slot = c->phiX + (size_t)c->part_rank * phix_chunk(c) + (size_t)c->bpr * hstep;
    HIPCHK(hipMemcpyAsync(slot, yN, hstep * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->bndY + (size_t)c->scan_blocks * hstep, yN, hstep * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->bndY2 + (size_t)c->scan_blocks2 * hstep, yN, hstep * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    return 0;
}

// adjoint, phases (ii)+(iii); phase (ii) two-level like the forward one
int qgdk_adjoint_finish(const qgdk_ctx *c)
{
    const size_t hstep = (size_t)c->Np * 2 * c->cp;
    const int B = c->scan_blocks, B2 = c->scan_blocks2, g = c->scan_g;
    const double *yN = c->phiX + (size_t)(c->part_world - 1) * phix_chunk(c) + (size_t)c->bpr * hstep;
    const double *PiRx = c->PiX + (size_t)c->bpr * 2 * c->Np * c->Np;      // panel copies inside the chunks
    int rc;
    const bool last = (c->part_rank == c->part_world - 1);
    if (!last) HIPCHK(hipMemcpyAsync(c->bndY + (size_t)B * hstep, yN, hstep * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    if (B2 <= 1) {
        ChainArgs s2{};
        s2.Np = c->Np; s2.cp = c->cp; s2.S = B; s2.Pmat = PiRx; s2.pm_bpr = c->bpr; s2.pm_chunk = (long long)pix_chunk(c);
        s2.start = c->bndY + (size_t)B * hstep; s2.start_stride = 0; s2.out = c->bndY;
        s2.forcing = c->phiX; s2.f_bpr = c->bpr; s2.nblocks = 1; s2.blen = B; s2.ngroups = c->cp / 8;
        if ((rc = launch_chain<3>(s2, c->stream))) return rc;
    } else {
        ChainArgs a2{};   // (ii-a) affine parts of the super-blocks (their propagators PiR2 come from the forward sweep)
        a2.Np = c->Np; a2.cp = c->cp; a2.S = B; a2.Pmat = PiRx; a2.pm_bpr = c->bpr; a2.pm_chunk = (long long)pix_chunk(c);
        a2.forcing = c->phiX; a2.f_bpr = c->bpr; a2.phi = c->phi2; a2.nblocks = B2; a2.blen = g; a2.ngroups = c->cp / 8;
        if ((rc = launch_chain<2>(a2, c->stream))) return rc;
        if (!last) HIPCHK(hipMemcpyAsync(c->bndY2 + (size_t)B2 * hstep, yN, hstep * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        ChainArgs b2{};   // (ii-b) y at super-block starts
        b2.Np = c->Np; b2.cp = c->cp; b2.S = B2; b2.Pmat = c->PiR2; b2.start = c->bndY2 + (size_t)B2 * hstep; b2.start_stride = 0;
        b2.out = c->bndY2; b2.forcing = c->phi2; b2.nblocks = 1; b2.blen = B2; b2.ngroups = c->cp / 8;
        if ((rc = launch_chain<3>(b2, c->stream))) return rc;
        ChainArgs c2{};   // (ii-c) y at every block start
        c2.Np = c->Np; c2.cp = c->cp; c2.S = B; c2.Pmat = PiRx; c2.pm_bpr = c->bpr; c2.pm_chunk = (long long)pix_chunk(c);
        c2.start = c->bndY2 + hstep; c2.start_stride = (long long)hstep; c2.out = c->bndY; c2.forcing = c->phiX; c2.f_bpr = c->bpr;
        c2.nblocks = B2; c2.blen = g; c2.ngroups = c->cp / 8;
        if ((rc = launch_chain<3>(c2, c->stream))) return rc;
    }
    // y at the end of this rank's window
    if (!last)
        HIPCHK(hipMemcpyAsync(c->yhist + (size_t)(c->nt - 1) * hstep, c->bndY + (size_t)c->blk_hi_clamped * hstep,
                              hstep * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    ChainArgs s3{};
    s3.Np = c->Np; s3.cp = c->cp; s3.S = c->nt - 1; s3.Pmat = c->Pr; s3.start = c->bndY + (size_t)(c->blk_lo + 1) * hstep;
    s3.start_stride = (long long)hstep; s3.out = c->yhist; s3.forcing = c->forcing; s3.nblocks = c->blk_hi - c->blk_lo;
    s3.blen = c->scan_blen; s3.ngroups = c->cp / 8;
    return launch_chain<3>(s3, c->stream);
}

int qgdk_lambda(const qgdk_ctx *c)
{
    size_t shm = (size_t)c->Np * 16 * sizeof(double);
    hipLaunchKernelGGL(k_lambda, dim3(c->cp / 8, c->nt - 1), dim3(256), shm, c->stream, c->LinvT, c->yhist, c->lam,
                       c->Np, c->cp, c->sigma, c->nt * c->n_ops * c->m * 2, c->grad, c->n_pcof);
    return (int)hipGetLastError();
}

int qgdk_derivs(const qgdk_ctx *c)
{
    size_t shm = (size_t)(c->m + 1) * c->Np * 16 * sizeof(double);
    double *gp = nullptr;
    if (c->panel_scratch) { gp = c->panel_scratch; shm = 0; }
#define CALL_DV(N) do { if (shm > 64 * 1024) HIPCHK(hipFuncSetAttribute((const void *)k_derivs<N>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm)); \
        hipLaunchKernelGGL((k_derivs<N>), dim3(c->cp / 8, c->nt), dim3(256), shm, c->stream, c->ops, c->tab, c->hist, c->dpsi, \
                           c->Np, c->cp, c->n_ops, c->m, gp); } while (0)
    DISPATCH_NOPS(c->n_ops, CALL_DV)
#undef CALL_DV
    return (int)hipGetLastError();
}

int qgdk_gradient(const qgdk_ctx *c)
{
    size_t shm = ((size_t)2 * c->m * c->Np * 16 + (size_t)c->n_ops * c->m * 2) * sizeof(double);
    double *gp = nullptr;
    if (c->panel_scratch) { gp = c->panel_scratch; shm = (size_t)c->n_ops * c->m * 2 * sizeof(double); }
    if (c->Np == 64 && c->m <= 5 && c->n_ops >= 1) {
        int rc = 0;
        switch (c->m) {
        case 1: rc = launch_gradpoint64_m<1>(c); break;
        case 2: rc = launch_gradpoint64_m<2>(c); break;
        case 3: rc = launch_gradpoint64_m<3>(c); break;
        case 4: rc = launch_gradpoint64_m<4>(c); break;
        default: rc = launch_gradpoint64_m<5>(c); break;
        }
        if (rc) return rc;
        return qgdk_contract(c);
    }
#define CALL_GS(N) do { if (shm > 64 * 1024) HIPCHK(hipFuncSetAttribute((const void *)k_gradsweep<N>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm)); \
        hipLaunchKernelGGL((k_gradsweep<N>), dim3(c->cp / 8, c->nt), dim3(256), shm, c->stream, c->ops, c->tab, c->hist, \
                           c->dpsi, c->lam, c->sigma, c->cw, c->Np, c->cp, c->n_ops, c->m, c->nt, gp); } while (0)
    DISPATCH_NOPS(c->n_ops, CALL_GS)
#undef CALL_GS
    return qgdk_contract(c);
}

int qgdk_contract(const qgdk_ctx *c)
{
    hipLaunchKernelGGL(k_contract, dim3((c->nt + CT_CHUNK - 1) / CT_CHUNK, c->n_ops, (c->nc_max + 63) / 64), dim3(256), 0,
                       c->stream, c->G, c->goff, c->ncoef, c->poff, c->sigma, c->grad, c->nt, c->m, c->n_ops);
    return (int)hipGetLastError();
}

int qgdk_apply(const qgdk_ctx *c, const double *in, double *out, int n, int d, double sign)
{
    size_t shm = (size_t)c->Np * 16 * sizeof(double);
#define CALL_AP(N) hipLaunchKernelGGL((k_apply<N>), dim3(c->cp / 8), dim3(256), shm, c->stream, c->ops, c->tab, in, out, c->Np, \
                                      c->cp, c->n_ops, c->m, n, d, sign)
    DISPATCH_NOPS(c->n_ops, CALL_AP)
#undef CALL_AP
    return (int)hipGetLastError();
}

int qgdk_gradient_needs_derivs(const qgdk_ctx *c) { return !(c->Np == 64 && c->m <= 5 && c->n_ops >= 1); }

// dynamic LDS of the chain/lambda kernels for any Np (they keep one or two panels only)
size_t qgdk_lds_needed(int Np, int m, int n_ops)
{
    size_t a = (size_t)(m + 1) * Np * 16 * sizeof(double);
    size_t b = ((size_t)2 * m * Np * 16 + (size_t)n_ops * m * 2) * sizeof(double);
    return a > b ? a : b;
}

} // extern "C"
