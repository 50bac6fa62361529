// qgd_k_build.hip -- control tables and the step matrices L_n, R_n
// (conventions and layouts: qgd_kernels_common.h; algorithm: DESIGN.md)
#include "qgd_kernels_common.h"
#include "qgd_front.h"
#include <string.h>

// The counters of the inverse (status[1]: matrices redone, status[2], N = 64: by the last resort) start at zero, and what the
// N = 64 kernel remembers between evaluations (status[3]: evaluations left that START with pivoting): when the diagonal-pivot
// attempt of the last evaluation was given up for more than a quarter of its matrices, the next QGD_PIVOT_FIRST_EVALS evaluations
// skip it; then it is tried again -- one evaluation with non-finite coefficients (a line search that overshoots) must not slow
// the handle down for good, and a problem whose matrices are not diagonally dominant pays for the failed attempt once in 33.
#define QGD_PIVOT_FIRST_EVALS 32
__device__ __forceinline__ void inverse_memory(int *status, int nt)
{
    const int left = status[3];
    if (left > 0) status[3] = left - 1;                       // (the last evaluation did not try the diagonal: nothing to learn from its count)
    else if (4 * status[1] > nt) status[3] = QGD_PIVOT_FIRST_EVALS;
    status[1] = 0; status[2] = 0;
}

// ---------------------------------------------------------------------------
// K0: control tables  tab[n][d][k][pq] = sum_l G[k][n][d][l] * pcof[off_k + l]
// (fill_p_mat!/fill_q_mat!, Control.jl:125-149, for the whole grid at once)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_tables(const double *__restrict__ G, const int64_t *__restrict__ goff,
                         const int32_t *__restrict__ ncoef, const int32_t *__restrict__ poff,
                         const double *__restrict__ pcof, double *__restrict__ tab, int nt, int m,
                         int n_ops, double *__restrict__ scal, int *__restrict__ status, int g_nt, int g_n0, int keep)
{
    // 16 lanes per table entry: contiguous 128-byte reads of the basis row, DPP sum over the 16 lanes.
    // The basis holds g_nt time points and this launch covers nt of them from g_n0 on (a window of a chunked grid;
    // g_nt = nt, g_n0 = 0 otherwise).  keep: a later chunk of the same evaluation -- the scalars go on accumulating.
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (!keep) {
        if (gid < 4) scal[gid] = 0.0;        // objective scalars and the singularity flag start at zero
        if (gid == 4) *status = 0;
        if (gid == 5) inverse_memory(status, nt);
    }
    const int idx = gid >> 4, sub = gid & 15;
    const int total = nt * (m + 1) * n_ops * 2;
    double s = 0.0;
    if (idx < total) {
        const int pq = idx & 1;
        const int k = (idx >> 1) % n_ops;
        const int d = ((idx >> 1) / n_ops) % (m + 1);
        const int n = ((idx >> 1) / n_ops) / (m + 1);
        const int nc = ncoef[k];
        // G for control k: [pq][nt][m+1][nc]
        const double *g = G + goff[k] + (((size_t)pq * g_nt + n + g_n0) * (m + 1) + d) * nc;
        const double *pc = pcof + poff[k];
        for (int l = sub; l < nc; l += 16) s = __builtin_fma(g[l], pc[l], s);
    }
    s = row16_sum(s);
    if (idx < total && sub == 15) tab[idx] = s;
}

// The same with the coefficient vector passed BY VALUE in the kernel arguments (up to QGD_PCOF_KERNARG doubles):
// an evaluation then starts with this kernel instead of with a 1.4 KB host-to-device copy packet and the gap
// behind it (2.9 + 6 us on the timeline of one cnot3 evaluation).
template <int NMAX> struct PcofArg { double v[NMAX]; };      // (the launch cost grows with the argument size: 3 sizes)
template <int NMAX>
__global__ __launch_bounds__(256) void k_tables_arg(const double *__restrict__ G, const int64_t *__restrict__ goff,
                         const int32_t *__restrict__ ncoef, const int32_t *__restrict__ poff,
                         const PcofArg<NMAX> pcof, double *__restrict__ tab, int nt, int m,
                         int n_ops, double *__restrict__ scal, int *__restrict__ status, int g_nt, int g_n0, int keep)
{
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (!keep) {
        if (gid < 4) scal[gid] = 0.0;
        if (gid == 4) *status = 0;
        if (gid == 5) inverse_memory(status, nt);
    }
    const int idx = gid >> 4, sub = gid & 15;
    const int total = nt * (m + 1) * n_ops * 2;
    double s = 0.0;
    if (idx < total) {
        const int pq = idx & 1;
        const int k = (idx >> 1) % n_ops;
        const int d = ((idx >> 1) / n_ops) % (m + 1);
        const int n = ((idx >> 1) / n_ops) / (m + 1);
        const int nc = ncoef[k];
        const double *g = G + goff[k] + (((size_t)pq * g_nt + n + g_n0) * (m + 1) + d) * nc;
        const double *pc = pcof.v + poff[k];
        for (int l = sub; l < nc; l += 16) s = __builtin_fma(g[l], pc[l], s);
    }
    s = row16_sum(s);
    if (idx < total && sub == 15) tab[idx] = s;
}

// The first launch of an evaluation on the fused-front path (qgd_front.h; Np = 64, sparse operators, pcof in the kernel
// arguments): the control tables as in k_tables_arg, and -- in the FIRST npre workgroups of the grid -- the step matrices
// L_n^H, R_n^H of the time points whose k_front workgroup should start with its elimination (front_pre_point).  A pre-building workgroup forms the (m+1) n_ops 2 table entries of its time point itself, with the
// arithmetic of the table workgroups (same bits: both store them).  1024 threads: a pre-building workgroup is alone on its CU,
// its four 16-column slabs go side by side (front_build<., ., 1024>); every workgroup carries the build's LDS (one per CU), so
// the table part is at most 256 - npre workgroups that stride over the entries.
#ifdef QGD_STAMPS      // scripts/build_variant.sh stamps -DQGD_STAMPS: device wall clock (10 ns) at the marks of the first 512 workgroups
__device__ unsigned long long g_stamps_build[512][8];
extern "C" int qgdk_stamps_build(unsigned long long *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps_build), sizeof(g_stamps_build)); }
#define TF_STAMP(i) do { if (blockIdx.x < 512 && threadIdx.x == 0) g_stamps_build[blockIdx.x][i] = wall_clock64(); } while (0)
#else
#define TF_STAMP(i) do { } while (0)
#endif
template <int NMAX, int M, int NOPS, int NTH>
__global__ __launch_bounds__(NTH) void k_tables_front(const double *__restrict__ G, const int64_t *__restrict__ goff,
                         const int32_t *__restrict__ ncoef, const int32_t *__restrict__ poff,
                         const PcofArg<NMAX> pcof, double *__restrict__ tab, const int nt, const int n_ops,
                         double *__restrict__ scal, int *__restrict__ status,
                         const FrontPre pre, const int32_t *__restrict__ ell_col, const uint8_t *__restrict__ ell_inv,
                         const double *__restrict__ ell_val, const double *__restrict__ cw, const int Z,
                         double *__restrict__ Eh, double *__restrict__ Fh)
{
    extern __shared__ __attribute__((aligned(32))) double front_smem[];
    const int per = (M + 1) * n_ops * 2, total = nt * per, sub = threadIdx.x & 15;
    auto entry = [&](const int idx) -> double {
        const int pq = idx & 1;
        const int k = (idx >> 1) % n_ops;
        const int d = ((idx >> 1) / n_ops) % (M + 1);
        const int n = ((idx >> 1) / n_ops) / (M + 1);
        const int nc = ncoef[k];
        const double *g = G + goff[k] + (((size_t)pq * nt + n) * (M + 1) + d) * nc;
        const double *pc = pcof.v + poff[k];
        double s = 0.0;
        // (the loads of four steps in flight together: the rolled loop waited for a memory round trip per 16 coefficients --
        //  at the head of a pre-building workgroup that is the head of the evaluation.  Same sums in the same order.)
        for (int l0 = sub; l0 < nc; l0 += 64) {
            double gv[4], pv[4];
            #pragma unroll
            for (int u = 0; u < 4; u++) { const int l = l0 + 16 * u; gv[u] = (l < nc) ? g[l] : 0.0; pv[u] = (l < nc) ? pc[l] : 0.0; }
            #pragma unroll
            for (int u = 0; u < 4; u++) if (l0 + 16 * u < nc) s = __builtin_fma(gv[u], pv[u], s);
        }
        return row16_sum(s);
    };
    TF_STAMP(0);
    const int npre = front_pre_count(pre);
    if ((int)blockIdx.x < npre) {
        const int n = front_pre_point(blockIdx.x, pre);
        for (int e = threadIdx.x >> 4; e < per; e += NTH / 16) {
            const double s = entry(n * per + e);
            if (sub == 15) tab[n * per + e] = s;
        }
        __syncthreads();
        TF_STAMP(1);
        constexpr size_t panel = 64 * 128;
        front_build<M, NOPS, NTH>(front_smem, ell_col, ell_inv, ell_val, tab, cw, n, n_ops, Z, Eh + (size_t)n * panel, Fh + (size_t)n * panel);
        TF_STAMP(2);
        return;
    }
    const int tb = blockIdx.x - npre, ntb = gridDim.x - npre;
    if (tb == 0) {
        if (threadIdx.x < 4) scal[threadIdx.x] = 0.0;
        if (threadIdx.x == 4) *status = 0;
        if (threadIdx.x == 5) inverse_memory(status, nt);
    }
    for (int idx = tb * (NTH / 16) + (threadIdx.x >> 4); idx < total; idx += ntb * (NTH / 16)) {
        const double s = entry(idx);
        if (sub == 15) tab[idx] = s;
    }
    TF_STAMP(1);
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

template <int M, int NOPS>
__global__ __launch_bounds__(512) void k_build_LR64(const double *__restrict__ ops,
                                                    const double *__restrict__ tab,
                                                    double *__restrict__ L, double *__restrict__ R,
                                                    int n_ops, const double *__restrict__ cw)
{
    constexpr int NP = 64, NGW = 4, SW = 16 * NGW;   // slab width in doubles
    extern __shared__ double smem[];
    double *Dbuf = smem;                               // [M-1][NP][SW]
    const int n = blockIdx.x >> 1, h = blockIdx.x & 1;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int c16 = lane & 15, kk = lane >> 4;
    const int rb = wave & 3, gh = wave >> 2;           // row block, pair of groups inside the slab
    const int arow = rb * 16 + c16;
    constexpr int PW = 2 * NP;
    const size_t panel = (size_t)NP * PW;

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

template <int NMAX>
static int launch_tables_arg(const qgdk_ctx *c, const double *pcof_host, int n_pcof)
{
    PcofArg<NMAX> arg;
    memcpy(arg.v, pcof_host, sizeof(double) * n_pcof);
    int total = c->nt * (c->m + 1) * c->n_ops * 2;
    hipLaunchKernelGGL((k_tables_arg<NMAX>), dim3((total * 16 + 255) / 256 + 1), dim3(256), 0, c->stream, c->G, c->goff, c->ncoef,
                       c->poff, arg, c->tab, c->nt, c->m, c->n_ops, c->scal, c->status, c->g_nt ? c->g_nt : c->nt, c->g_n0, c->keep_scal);
    return (int)hipGetLastError();
}

template <int NMAX, int M, int NOPS, int NTH>
static int launch_tables_front_n(const qgdk_ctx *c, const double *pcof_host, int n_pcof, const FrontPre pre, int ntb)
{
    PcofArg<NMAX> arg;
    memcpy(arg.v, pcof_host, sizeof(double) * n_pcof);
    const size_t shm = front_build_lds(M, c->ell_z, NTH / 256);
    SET_LDS_ONCE((k_tables_front<NMAX, M, NOPS, NTH>), shm);
    hipLaunchKernelGGL((k_tables_front<NMAX, M, NOPS, NTH>), dim3(front_pre_count(pre) + ntb), dim3(NTH), shm, c->stream, c->G, c->goff, c->ncoef, c->poff, arg,
                       c->tab, c->nt, c->n_ops, c->scal, c->status, pre, c->ell_col, c->ell_inv, c->ell_val, c->cw, c->ell_z,
                       c->L, c->R);
    return (int)hipGetLastError();
}

// One round of workgroups: the pre-building workgroups (at most 192, qgdk_front_pre_plan) go one per CU with 1024 threads (the
// four slabs side by side) beside at least 64 table workgroups.
template <int NMAX, int M, int NOPS>
static int launch_tables_front(const qgdk_ctx *c, const double *pcof_host, int n_pcof)
{
    FrontPre pre{0, 0, 0};
    pre.extra = qgdk_front_pre_plan(c, &pre.q2, &pre.q1);
    const int npre = front_pre_count(pre), total = c->nt * (c->m + 1) * c->n_ops * 2;
    int ntb = (total + 63) / 64; if (ntb > 256 - npre) ntb = 256 - npre;
    return launch_tables_front_n<NMAX, M, NOPS, 1024>(c, pcof_host, n_pcof, pre, ntb);
}

template <int NMAX>
static int launch_tables_front_m(const qgdk_ctx *c, const double *pcof_host, int n_pcof)
{
    if (c->m == 4 && c->n_ops == 3) return launch_tables_front<NMAX, 4, 3>(c, pcof_host, n_pcof);
    switch (c->m) {
    case 1: return launch_tables_front<NMAX, 1, -1>(c, pcof_host, n_pcof);
    case 2: return launch_tables_front<NMAX, 2, -1>(c, pcof_host, n_pcof);
    case 3: return launch_tables_front<NMAX, 3, -1>(c, pcof_host, n_pcof);
    case 4: return launch_tables_front<NMAX, 4, -1>(c, pcof_host, n_pcof);
    default: return (int)hipErrorInvalidValue;
    }
}

extern "C" {

int qgdk_tables(const qgdk_ctx *c, const double *pcof)
{
    int total = c->nt * (c->m + 1) * c->n_ops * 2;
    hipLaunchKernelGGL(k_tables, dim3((total * 16 + 255) / 256 + 1), dim3(256), 0, c->stream, c->G, c->goff, c->ncoef,
                       c->poff, pcof, c->tab, c->nt, c->m, c->n_ops, c->scal, c->status, c->g_nt ? c->g_nt : c->nt, c->g_n0, c->keep_scal);
    return (int)hipGetLastError();
}

int qgdk_tables_kernarg(const qgdk_ctx *c, const double *pcof_host, int n_pcof)
{
    if (n_pcof <= 64) return launch_tables_arg<64>(c, pcof_host, n_pcof);
    if (n_pcof <= 192) return launch_tables_arg<192>(c, pcof_host, n_pcof);
    if (n_pcof <= QGD_PCOF_KERNARG) return launch_tables_arg<QGD_PCOF_KERNARG>(c, pcof_host, n_pcof);
    return (int)hipErrorInvalidValue;
}

// the fused-front path's first launch (qgdk_front_supported; pcof in the kernel arguments, the basis covering exactly the grid)
int qgdk_tables_front(const qgdk_ctx *c, const double *pcof_host, int n_pcof)
{
    { int q2, q1; if (qgdk_front_pre_plan(c, &q2, &q1) == 0) return qgdk_tables_kernarg(c, pcof_host, n_pcof); }      // (no tail to balance: the plain tables kernel)
    if (n_pcof <= 64) return launch_tables_front_m<64>(c, pcof_host, n_pcof);
    if (n_pcof <= 192) return launch_tables_front_m<192>(c, pcof_host, n_pcof);
    if (n_pcof <= QGD_PCOF_KERNARG) return launch_tables_front_m<QGD_PCOF_KERNARG>(c, pcof_host, n_pcof);
    return (int)hipErrorInvalidValue;
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
    if (c->use_sparse) return qgdk_build_LR_sparse(c);
    if (c->dense_gemm) return qgdk_dense_build_LR(c);
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


} // extern "C"
