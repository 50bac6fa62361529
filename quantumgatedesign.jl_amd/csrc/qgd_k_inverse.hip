// qgd_k_inverse.hip -- batched inverses of L_n and the step propagators
// (conventions and layouts: qgd_kernels_common.h; algorithm: DESIGN.md)
#include "qgd_kernels_common.h"

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
                                                 int use_lds, int *__restrict__ status, const int *__restrict__ redo = nullptr)
{
    extern __shared__ double smem[];
    if (redo) {                                      // fallback pass of the block Gauss-Jordan inverse (sizes beyond k_inverse_blocked2's LDS)
        if (!redo[n0 + blockIdx.x]) return;
        if (threadIdx.x == 0) atomicAdd(status + 1, 1);
    }
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
// K2 (fast path, Np <= 64): blocked Gauss-Jordan inverse, 4 pivots per panel, rank-4 updates on
// the fp64 MFMA.  The matrix lives in registers in accumulator layout (wave w owns rows
// 16w..16w+15 of all columns: d4 M[Np/8]) and rows are never moved: pivoting is implicit (the
// p-th pivot row rho(p) stays where it is) and the row/column permutation
// A^-1[i][rho(j)] = M[rho(i)][j] is applied when the result is written.  Per panel:
//   1. the 4 panel columns go to LDS;
//   2. wave 0 (lane = row) runs the pivoted in-place Gauss-Jordan steps on the Np x 4 panel only
//      (pivot search = DPP max-scan of a packed |x|^2/row key, pivot-row broadcast =
//      v_readlane): that yields the pivot rows, and -F*B^-1 (rows off the pivot block) /
//      B^-1 (pivot block), i.e. exactly the multipliers of the rank-4 block step and the
//      in-place inverse entries;
//   3. the owners of the 4 pivot rows publish them: they already are an MFMA B operand;
//   4. every wave: M += A * M[P,:] with A = multipliers (minus identity on the pivot rows) --
//      2 MFMAs per 16x8-complex tile -- then the pivot columns are overwritten with the multipliers.
// Pivot = largest modulus among the unused rows (compared on the upper 26 bits of |x|^2), the
// panel columns being fully updated when their pivots are chosen.  3 barriers per panel.
// ---------------------------------------------------------------------------
__device__ __forceinline__ double lane_read(double v, int l)     // l wave-uniform
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

template <int CTRL, int ROWMASK>
__device__ __forceinline__ unsigned dpp_max_step(unsigned key)
{
    const unsigned o = (unsigned)__builtin_amdgcn_update_dpp((int)key, (int)key, CTRL, ROWMASK, 0xF, false);
    return o > key ? o : key;
}

// maximum of a 32-bit key over the wave, returned wave-uniform
__device__ __forceinline__ unsigned wave_max_u32(unsigned key)
{
    key = dpp_max_step<0x111, 0xF>(key);      // row_shr:1
    key = dpp_max_step<0x112, 0xF>(key);      // row_shr:2
    key = dpp_max_step<0x114, 0xF>(key);      // row_shr:4
    key = dpp_max_step<0x118, 0xF>(key);      // row_shr:8   -> lane 15 of each row holds the row maximum
    key = dpp_max_step<0x142, 0xA>(key);      // row_bcast:15 into rows 1 and 3
    key = dpp_max_step<0x143, 0xC>(key);      // row_bcast:31 into rows 2 and 3
    return (unsigned)__builtin_amdgcn_readlane((int)key, 63);
}

__device__ __forceinline__ double fast_rcp(double d)             // v_rcp_f64 + 2 Newton steps (full precision)
{
    double r = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, r, 1.0); r = __builtin_fma(r, e, r);
    e = __builtin_fma(-d, r, 1.0); r = __builtin_fma(r, e, r);
    return r;
}

// LDS of the Np <= 64 kernels (doubles): the working buffers of the elimination, overlaid by the output staging plane
#define INV_SMEM(NP) (((8 * 2 * (NP) + 16 * (NP) + 8 * (NP)) > (NP) * ((NP) + 1)) ? (8 * 2 * (NP) + 16 * (NP) + 8 * (NP)) : (NP) * ((NP) + 1))

template <int NP>
__device__ __forceinline__ void inverse_mfma_body(const double *__restrict__ L, const double *__restrict__ R, double *__restrict__ LinvT,
                                                  double *__restrict__ Pr, double *__restrict__ Pc, const int n, int *__restrict__ status,
                                                  double *__restrict__ smem, int *__restrict__ rho, int *__restrict__ rinv)
{
    constexpr int NG = NP / 8, NW = NP / 16, NTH = 64 * NW, PW = 2 * NP, LDP = NP + 1;
    // LDS: the working buffers of the elimination, overlaid by the output staging plane at the end
    constexpr int O_PROW = 0, O_G = O_PROW + 8 * PW, O_F = O_G + 16 * NP,
                  WORK = O_F + 8 * NP, SM = (WORK > NP * LDP) ? WORK : NP * LDP;
    static_assert(SM <= INV_SMEM(NP), "INV_SMEM");
    double *Prow = smem + O_PROW;                       // [2][4][PW]   pivot rows (B operand), by panel parity
    double *Gm = smem + O_G;                            // [2][2][NP][4] multipliers re/im, by panel parity
    double *Fm = smem + O_F;                            // [2][NP][4]   panel columns re/im
    const int t = threadIdx.x, w = t >> 6, lane = t & 63;
    const int c16 = lane & 15, kk = lane >> 4;
    const size_t panel = (size_t)NP * PW, pl = (size_t)NP * NP;
    const double *Ln = L + (size_t)n * panel;

    d4 M[NG];
    #pragma unroll
    for (int g = 0; g < NG; g++)
        #pragma unroll
        for (int r = 0; r < 4; r++) M[g][r] = Ln[(size_t)(16 * w + kk + 4 * r) * PW + 16 * g + c16];
    bool used = lane >= NP;                             // panel wave: this lane's row has been a pivot row
    // which wave factors the panels (rotating it differently across workgroups that share a CU, e.g. with
    // blockIdx/256, changes nothing: 108-111 us for 550 matrices either way)
    const int pw = (NW > 1) ? (int)(blockIdx.x % NW) : 0;

    for (int pn = 0; pn < NP / 4; pn++) {
        const int p0 = pn * 4, gp = p0 >> 3, q0 = p0 & 7, par = pn & 1;
        double *Gre = Gm + par * 8 * NP, *Gim = Gre + 4 * NP;
        double *Pr = Prow + par * 4 * PW;
        // ---- 1. publish the panel columns
        {
            const int s = (c16 & 7) - q0;
            if (s >= 0 && s < 4) {
                double *dst = Fm + (c16 < 8 ? 0 : 4 * NP);
                #pragma unroll
                for (int g = 0; g < NG; g++) {
                    if (g == gp) {
                        #pragma unroll
                        for (int r = 0; r < 4; r++) dst[(16 * w + kk + 4 * r) * 4 + s] = M[g][r];
                    }
                }
            }
        }
        __syncthreads();
        // ---- 2. one wave (a different one in neighbouring workgroups, so that the serial phases of
        //         the workgroups sharing a CU sit on different SIMDs): pivoted in-place Gauss-Jordan
        //         on the NP x 4 panel, lane = row
        if (w == pw) {
            __builtin_amdgcn_s_setprio(1);      // the pivot chain is the critical path: ahead of other workgroups' MFMA bursts on this SIMD
            double xr[4], xi[4];
            const int lrow = (lane < NP) ? lane : 0;
            #pragma unroll
            for (int s = 0; s < 4; s++) { xr[s] = Fm[lrow * 4 + s]; xi[s] = Fm[4 * NP + lrow * 4 + s]; }
            #pragma unroll
            for (int s = 0; s < 4; s++) {
                const double m2 = xr[s] * xr[s] + xi[s] * xi[s];
                const unsigned mag = (unsigned)__double2hiint(m2);
                unsigned key = used ? 0u : ((mag & ~63u) | (unsigned)(63 - lane));
                key = wave_max_u32(key);
                const int pr = 63 - (int)(key & 63u);
                if (lane == 0) { rho[p0 + s] = pr; rinv[pr] = p0 + s; if ((key >> 6) == 0) *status = 1; }
                used = used || (lane == pr);
                double yr[4], yi[4];
                #pragma unroll
                for (int q = 0; q < 4; q++) { yr[q] = lane_read(xr[q], pr); yi[q] = lane_read(xi[q], pr); }
                const double den = fast_rcp(yr[s] * yr[s] + yi[s] * yi[s]);
                const double ir = yr[s] * den, ii = -yi[s] * den;
                const double fr = xr[s], fi = xi[s];
                #pragma unroll
                for (int q = 0; q < 4; q++) {
                    const double rr = (q == s) ? ir : yr[q] * ir - yi[q] * ii;      // scaled pivot row
                    const double ri = (q == s) ? ii : yr[q] * ii + yi[q] * ir;
                    const double br = (q == s) ? 0.0 : xr[q], bi = (q == s) ? 0.0 : xi[q];
                    xr[q] = (lane == pr) ? rr : br - (fr * rr - fi * ri);
                    xi[q] = (lane == pr) ? ri : bi - (fr * ri + fi * rr);
                }
            }
            if (lane < NP) {
                #pragma unroll
                for (int s = 0; s < 4; s++) { Gre[lane * 4 + s] = xr[s]; Gim[lane * 4 + s] = xi[s]; }
            }
            __builtin_amdgcn_s_setprio(0);
        }
        __syncthreads();
        // ---- 3. the owners of the pivot rows publish them as the B operand of the block step
        {
            const int s0 = rho[p0], s1 = rho[p0 + 1], s2 = rho[p0 + 2], s3 = rho[p0 + 3];
            #pragma unroll
            for (int r = 0; r < 4; r++) {
                const int x = 16 * w + kk + 4 * r;
                const int ps = (x == s0) ? 0 : (x == s1) ? 1 : (x == s2) ? 2 : (x == s3) ? 3 : -1;
                if (ps >= 0) {
                    #pragma unroll
                    for (int g = 0; g < NG; g++) Pr[ps * PW + 16 * g + c16] = M[g][r];
                }
            }
        }
        __syncthreads();
        // ---- 4. rank-4 block step on the MFMA, then the pivot columns take the multipliers
        {
            const int arow = 16 * w + c16;
            const double are = Gre[arow * 4 + kk] - ((arow == rho[p0 + kk]) ? 1.0 : 0.0);
            const double aim = Gim[arow * 4 + kk];
            {
            #pragma unroll
            for (int g = 0; g < NG; g++) {
                double b1, b2;
                panel_b(Pr + kk * PW + 16 * g, c16, b1, b2);
                M[g] = MFMA(are, b1, M[g]);
                M[g] = MFMA(aim, b2, M[g]);
            }
            }
            const int s = (c16 & 7) - q0;
            if (s >= 0 && s < 4) {
                const double *src = (c16 < 8) ? Gre : Gim;
                #pragma unroll
                for (int g = 0; g < NG; g++) {
                    if (g == gp) {
                        #pragma unroll
                        for (int r = 0; r < 4; r++) M[g][r] = src[(16 * w + kk + 4 * r) * 4 + s];
                    }
                }
            }
        }
        // no barrier here: the next panel writes F (last read before barrier 2), and the buffers
        // read above (G, Prow) alternate with the panel parity
    }
    __syncthreads();
    // (from here on the barriers order LDS traffic only -- lds_barrier: __syncthreads would also wait for the global
    //  stores of the previous plane to be acknowledged, four times)
    // ---- output.  A^-1[rinv[x]][rho[j]] = M[x][j] goes through LDS one plane at a time (real parts,
    // then imaginary parts): each staged plane is written out as LinvT (left operand of
    // lambda = L^-H y) and is at once the A operand of one half of the step propagator
    //   P_{n-1} = L_n^-1 R_{n-1} = Are [Rre|Rim] + Aim [-Rim|Rre]
    // (forward_evolution.jl:181-220, the implicit solve done for all right-hand sides).  For the
    // product wave w owns the column groups 2w, 2w+1 of P over all rows: R is read once.
    double *T = LinvT + (size_t)n * 2 * pl;
    const double *Rn = R + (size_t)(n - 1) * panel;
    int orow[4];
    #pragma unroll
    for (int r = 0; r < 4; r++) orow[r] = rinv[16 * w + kk + 4 * r] * LDP;
    constexpr int NRT = NP / 16, GPW = NG / NW;            // row tiles, column groups per wave
    d4 acc[NRT][GPW];
    #pragma unroll
    for (int rt = 0; rt < NRT; rt++)
        #pragma unroll
        for (int gg = 0; gg < GPW; gg++) acc[rt][gg] = (d4){0, 0, 0, 0};
    #pragma unroll
    for (int pass = 0; pass < 2; pass++) {
        if ((c16 >> 3) == pass) {
            #pragma unroll
            for (int g = 0; g < NG; g++) {
                const int oc = rho[8 * g + (c16 & 7)];
                #pragma unroll
                for (int r = 0; r < 4; r++) smem[orow[r] + oc] = M[g][r];
            }
        }
        lds_barrier();
        {   // (buffer-addressed: descriptor and the constant part of the offset in SGPRs, no 64-bit vector adds)
            const __amdgpu_buffer_rsrc_t rT = buffer_of(T + pass * pl);
            #pragma unroll
            for (int q = 0; q < NP * NP / NTH; q++) {
                const int e = t + q * NTH;
                buffer_store_f64(smem[(e / NP) * LDP + (e % NP)], rT, t * 8, q * NTH * 8);
            }
        }
        #pragma unroll 4
        for (int ks = 0; ks < NP / 4; ks++) {
            const int k = 4 * ks + kk;
            double bf[GPW];
            #pragma unroll
            for (int gg = 0; gg < GPW; gg++) {
                const double *row = Rn + (size_t)k * PW + 16 * (GPW * w + gg);
                if (pass == 0) bf[gg] = row[c16];
                else { const double v = row[c16 ^ 8]; bf[gg] = (c16 < 8) ? -v : v; }
            }
            #pragma unroll
            for (int rt = 0; rt < NRT; rt++) {
                const double af = smem[(16 * rt + c16) * LDP + k];
                #pragma unroll
                for (int gg = 0; gg < GPW; gg++) acc[rt][gg] = MFMA(af, bf[gg], acc[rt][gg]);
            }
        }
        lds_barrier();
    }
    // P: panel (row-major, left operand of the adjoint sweep as P^H) straight from the accumulators,
    // column-major planes (left operand of the forward sweep) through LDS
    double *Prn = Pr + (size_t)(n - 1) * panel, *Pcn = Pc + (size_t)(n - 1) * 2 * pl;
    #pragma unroll
    for (int rt = 0; rt < NRT; rt++)
        #pragma unroll
        for (int gg = 0; gg < GPW; gg++)
            #pragma unroll
            for (int r = 0; r < 4; r++)
                Prn[(size_t)(16 * rt + kk + 4 * r) * PW + 16 * (GPW * w + gg) + c16] = acc[rt][gg][r];
    #pragma unroll
    for (int pass = 0; pass < 2; pass++) {
        if ((c16 >> 3) == pass) {
            #pragma unroll
            for (int rt = 0; rt < NRT; rt++)
                #pragma unroll
                for (int gg = 0; gg < GPW; gg++)
                    #pragma unroll
                    for (int r = 0; r < 4; r++)
                        smem[(8 * (GPW * w + gg) + (c16 & 7)) * LDP + 16 * rt + kk + 4 * r] = acc[rt][gg][r];   // [col][row]
        }
        lds_barrier();
        {
            const __amdgpu_buffer_rsrc_t rC = buffer_of(Pcn + pass * pl);
            #pragma unroll
            for (int q = 0; q < NP * NP / NTH; q++) {
                const int e = t + q * NTH;
                buffer_store_f64(smem[(e / NP) * LDP + (e % NP)], rC, t * 8, q * NTH * 8);
            }
        }
        lds_barrier();
    }
}

template <int NP>
__global__ __launch_bounds__(NP * 4) __attribute__((amdgpu_waves_per_eu(3, 3)))
void k_inverse_mfma(const double *__restrict__ L, const double *__restrict__ R, double *__restrict__ LinvT,
                    double *__restrict__ Pr, double *__restrict__ Pc, int n0, int *__restrict__ status)
{
    __shared__ double smem[INV_SMEM(NP)];
    __shared__ int rho[NP], rinv[NP];                   // rho[p] = row of the p-th pivot
    inverse_mfma_body<NP>(L, R, LinvT, Pr, Pc, n0 + (int)blockIdx.x, status, smem, rho, rinv);
}

#include "qgd_inverse_cb.h"

// The last resort of the column-block kernel: the fully pivoted elimination above on the same matrix, as a function that
// ends the wave.
__device__ __attribute__((noinline, noreturn)) void cb_fallback(const double *L, const double *R, double *LinvT, double *Pr, double *Pc, int n,
                                                                int *status, int *fallbacks, double *smem, int *rho, int *rinv, int front)
{
    if (fallbacks && threadIdx.x == 0) atomicAdd(fallbacks + 1, 1);             // (matrices both column-block attempts gave up)
    inverse_mfma_body<64>(cb_uniform(L), cb_uniform(R), cb_uniform(LinvT), cb_uniform(Pr), cb_uniform(Pc), __builtin_amdgcn_readfirstlane(n),
                          cb_uniform(status), cb_lds(smem), cb_lds(rho), cb_lds(rinv));
    if (front) front_relayout(L, R, Pr, Pc, n);
    __builtin_amdgcn_endpgm();
}

// The pivoted stages leave Y = L^-1 R as a row-major panel (Pr[n-1]) and as column-major planes (Pc[n-1]); the fused front's
// sweeps read S = Y^H as column-major planes from Pc and as a row-major panel from Pr (qgd_front.h).  Both are row-local
// re-interleavings: planes[p][a][b] = +-panel[a][b]_p and panel[a][b]_p = +-planes[p][a][b].  The first goes through the panel
// of R[n-1] (an input, dead by now; L[n] stays as it is: phi_0 is formed from L_0^H later), the second straight across.
// 256 threads, the whole workgroup.
__device__ __attribute__((noinline)) void front_relayout(const double *L, const double *R, double *Pr, double *Pc, int n)
{
    constexpr int NP = 64, PW = 128;
    const size_t panel = (size_t)NP * PW;
    (void)L;
    double *F = const_cast<double *>(R) + (size_t)(n - 1) * panel;
    double *pr = Pr + (size_t)(n - 1) * panel, *pc = Pc + (size_t)(n - 1) * panel;
    __syncthreads();                                    // (the stages' own stores have reached memory)
    for (int idx = threadIdx.x; idx < NP * PW; idx += 256) {
        const int p = idx >> 12, a = (idx >> 6) & 63, b = idx & 63, pan = a * PW + (b >> 3) * 16 + (b & 7) + 8 * p;
        const double v = __builtin_nontemporal_load(pr + pan);
        F[idx] = p ? -v : v;                            // conj(Y) in row-major planes (to become Pc)
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < NP * PW; idx += 256) {
        const int p = idx >> 12, a = (idx >> 6) & 63, b = idx & 63, pan = a * PW + (b >> 3) * 16 + (b & 7) + 8 * p;
        const double v = __builtin_nontemporal_load(pc + idx);
        pr[pan] = p ? -v : v;                           // conj(Y) as a column-major panel
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < NP * PW; idx += 256) pc[idx] = F[idx];
}

// Np = 64: column-block elimination first; a matrix whose diagonal tiles do not carry the pivots (zero pivot or a multiplier
// beyond CB_GROWTH inside a tile) is done again by the fully pivoted elimination above, in the same workgroup.
// fallbacks: three words or null (inverse_cb_body).  ONE: every workgroup of the launch is resident at once (qgd_inverse_cb.h).
template <bool ONE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3)))
void k_inverse_cb(const double *__restrict__ L, const double *__restrict__ R, double *__restrict__ LinvT,
                  double *__restrict__ Pr, double *__restrict__ Pc, int n0, int *__restrict__ status, int *__restrict__ fallbacks)
{
    constexpr int SM = (INV_SMEM(64) > CB_WORK) ? INV_SMEM(64) : CB_WORK;
    static_assert(SM >= 4 * 16 * CB_LDP, "staging slabs of the P planes");
    __shared__ __attribute__((aligned(32))) double smem[SM];
    __shared__ int rho[64], rinv[64];
    __shared__ int bad;
    inverse_cb_body<ONE>(L, R, LinvT, Pr, Pc, n0 + (int)blockIdx.x, status, fallbacks, smem, rho, rinv, &bad);
}

#include "qgd_front.h"
#ifndef FRONT_PRIO_BUILD
#define FRONT_PRIO_BUILD 3
#endif
#ifdef CB_PROFILE       // scripts/ubench/front_bench.hip -DCB_PROFILE: s_memtime at the start and end of the build phase of every workgroup
__device__ unsigned long long g_front_prof[4096][4];      // wall clock (100 MHz, one counter for the whole device): start, build end, elimination start, end
#define FRONT_STAMP(i) do { if (blockIdx.x < 4096 && threadIdx.x == 0) g_front_prof[blockIdx.x][i] = wall_clock64(); } while (0)
#else
#define FRONT_STAMP(i) do { } while (0)
#endif

// The fused front (qgd_front.h): workgroup n builds L_n^H, R_n^H into Eh[n], Fh[n] and eliminates [L_n^H | R_n^H] in place of
// k_build_LR_ell + k_inverse_cb.  LinvT[n] = L_n^-H (row-major planes); Pc[n] / Pr[n] = S_n = R_n L_n^-1 as column-major planes /
// row-major panel -- what the sweeps read where the two-point form has P_n -- for n = 0 .. nt-1.  pre: the time points
// front_is_prebuilt(n, pre) were built by the launch in front (k_tables_front / k_front_pre); all zero: every workgroup builds its own.
// Dynamic LDS: max(front_build_lds, the column-block kernel's work space).
template <int M, int NOPS, bool ONE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3)))
void k_front(const int32_t *__restrict__ ell_col, const uint8_t *__restrict__ ell_inv, const double *__restrict__ ell_val,
             const double *__restrict__ tab, const double *__restrict__ cw, int n_ops, int Z,
             double *__restrict__ Eh, double *__restrict__ Fh, double *__restrict__ LinvT, double *__restrict__ Pr,
             double *__restrict__ Pc, int *__restrict__ status, int *__restrict__ fallbacks, const FrontPre pre)
{
    extern __shared__ __attribute__((aligned(32))) double front_smem[];
    __shared__ int rho[64], rinv[64];
    __shared__ int bad;
    constexpr size_t panel = 64 * 128, pl = 64 * 64;
    const int n = blockIdx.x;
    FRONT_STAMP(0);
    if (!front_is_prebuilt(n, pre)) {
        // (the build ahead of everything else on the CU: a workgroup that is still building while its neighbours eliminate at
        //  raised priority crawls -- 113 -> 109 us for the 551 time points of the headline)
        __builtin_amdgcn_s_setprio(FRONT_PRIO_BUILD);
        front_build<M, NOPS>(front_smem, ell_col, ell_inv, ell_val, tab, cw, n, n_ops, Z, Eh + (size_t)n * panel, Fh + (size_t)n * panel);
        __builtin_amdgcn_s_setprio(0);
    }
    FRONT_STAMP(1);
    __syncthreads();        // (s_waitcnt vmcnt(0) + s_barrier: the panels are in the L2 every wave of this workgroup reads through)
    FRONT_STAMP(2);
    // the column-block kernel pairs L[n] with R[n-1], Pr[n-1], Pc[n-1]: shifted by one matrix they are this time point's
    inverse_cb_body<ONE, true>(Eh, Fh + panel, LinvT, Pr + panel, Pc + 2 * pl, n, status, fallbacks, front_smem, rho, rinv, &bad);
    FRONT_STAMP(3);
}

// The pre-built step matrices as a launch of their own (scripts/ubench/front_bench.hip; the library builds them in k_tables_front).
template <int M, int NOPS>
__global__ __launch_bounds__(256) void k_front_pre(const int32_t *__restrict__ ell_col, const uint8_t *__restrict__ ell_inv, const double *__restrict__ ell_val,
                                                   const double *__restrict__ tab, const double *__restrict__ cw, int n_ops, int Z,
                                                   double *__restrict__ Eh, double *__restrict__ Fh, const FrontPre pre)
{
    extern __shared__ __attribute__((aligned(32))) double front_smem[];
    constexpr size_t panel = 64 * 128;
    const int n = front_pre_point(blockIdx.x, pre);
    front_build<M, NOPS>(front_smem, ell_col, ell_inv, ell_val, tab, cw, n, n_ops, Z, Eh + (size_t)n * panel, Fh + (size_t)n * panel);
}
static inline size_t front_lds(int M, int Z)
{
    constexpr size_t SM = ((INV_SMEM(64) > CB_WORK) ? INV_SMEM(64) : CB_WORK) * sizeof(double);
    const size_t b = front_build_lds(M, Z);
    return b > SM ? b : SM;
}

// ---------------------------------------------------------------------------
// The elimination of k_inverse_mfma as a function of a matrix held in registers (used by k_inverse_diag for the diagonal
// blocks of the N > 64 block inverse, qgd_k_dense.hip).  STATIC = true takes the diagonal as pivots and reports a multiplier
// beyond a modulus of 8 instead of searching (measured in round 2 as a first attempt for whole matrices: no gain inside the
// evaluation); the library instantiates the pivoted form only.
// ---------------------------------------------------------------------------
#define INV_GROWTH2 64.0
#define INVM_PROF(i) do { } while (0)
#define INVM_PROF_DECL
#define INVM_PROF_ARG

template <int NP, bool STATIC>
__device__ __forceinline__ bool gj_panels(INVM_PROF_DECL d4 (&M)[NP / 8], double *__restrict__ work, int *__restrict__ rho,
                                          int *__restrict__ rinv, const int w, const int lane, const int pw,
                                          int *__restrict__ status)
{
    constexpr int NG = NP / 8, PW = 2 * NP;
    constexpr int O_PROW = 0, O_G = O_PROW + 8 * PW, O_F = O_G + 16 * NP;
    double *Prow = work + O_PROW;                       // [2][4][PW]   pivot rows (B operand), by panel parity
    double *Gm = work + O_G;                            // [2][2][NP][4] multipliers re/im, by panel parity
    double *Fm = work + O_F;                            // [2][NP][4]   panel columns re/im
    const int c16 = lane & 15, kk = lane >> 4;
    bool used = lane >= NP;                             // panel wave, pivoted pass: this lane's row has been a pivot row
    bool bad = false;
    for (int pn = 0; pn < NP / 4; pn++) {
        const int p0 = pn * 4, gp = p0 >> 3, q0 = p0 & 7, par = pn & 1;
        double *Gre = Gm + par * 8 * NP, *Gim = Gre + 4 * NP;
        double *Pr = Prow + par * 4 * PW;
        // ---- 1. publish the panel columns; static pass: and the pivot rows p0..p0+3 (rows 16w+kk+4r: wave p0/16,
        //         register (p0/4)%4, lanes kk = 0..3)
        {
            const int s = (c16 & 7) - q0;
            if (s >= 0 && s < 4) {
                double *dst = Fm + (c16 < 8 ? 0 : 4 * NP);
                #pragma unroll
                for (int g = 0; g < NG; g++) {
                    if (g == gp) {
                        #pragma unroll
                        for (int r = 0; r < 4; r++) dst[(16 * w + kk + 4 * r) * 4 + s] = M[g][r];
                    }
                }
            }
        }
        if (STATIC && w == (p0 >> 4)) {
            const int rsel = (p0 >> 2) & 3;
            #pragma unroll
            for (int r = 0; r < 4; r++) {
                if (r == rsel) {
                    #pragma unroll
                    for (int g = 0; g < NG; g++) Pr[kk * PW + 16 * g + c16] = M[g][r];
                }
            }
        }
        INVM_PROF(1);
        lds_barrier();
        INVM_PROF(2);
        // ---- 2. the panel wave of this matrix: in-place Gauss-Jordan on the NP x 4 panel, lane = row
        if (w == pw) {
            double xr[4], xi[4];
            const int lrow = (lane < NP) ? lane : 0;
            #pragma unroll
            for (int s = 0; s < 4; s++) { xr[s] = Fm[lrow * 4 + s]; xi[s] = Fm[4 * NP + lrow * 4 + s]; }
            #pragma unroll
            for (int s = 0; s < 4; s++) {
                int pr;
                if (STATIC) pr = p0 + s;
                else {
                    const unsigned mag = (unsigned)__double2hiint(xr[s] * xr[s] + xi[s] * xi[s]);
                    unsigned key = used ? 0u : ((mag & ~63u) | (unsigned)(63 - lane));
                    key = wave_max_u32(key);
                    pr = 63 - (int)(key & 63u);
                    if (lane == 0) { rho[p0 + s] = pr; rinv[pr] = p0 + s; if ((key >> 6) == 0) *status = 1; }
                    used = used || (lane == pr);
                }
                double yr[4], yi[4];
                #pragma unroll
                for (int q = 0; q < 4; q++) { yr[q] = lane_read(xr[q], pr); yi[q] = lane_read(xi[q], pr); }
                const double den = fast_rcp(yr[s] * yr[s] + yi[s] * yi[s]);
                const double ir = yr[s] * den, ii = -yi[s] * den;
                const double fr = xr[s], fi = xi[s];
                if (STATIC) {      // |multiplier|^2 of every other row; NaN (zero pivot) counts as too large
                    const double m2 = (fr * fr + fi * fi) * den;
                    bad = bad || (lane != pr && lane < NP && !(m2 <= INV_GROWTH2));
                }
                #pragma unroll
                for (int q = 0; q < 4; q++) {
                    const double rr = (q == s) ? ir : yr[q] * ir - yi[q] * ii;      // scaled pivot row
                    const double ri = (q == s) ? ii : yr[q] * ii + yi[q] * ir;
                    const double br = (q == s) ? 0.0 : xr[q], bi = (q == s) ? 0.0 : xi[q];
                    xr[q] = (lane == pr) ? rr : br - (fr * rr - fi * ri);
                    xi[q] = (lane == pr) ? ri : bi - (fr * ri + fi * rr);
                }
            }
            if (lane < NP) {
                #pragma unroll
                for (int s = 0; s < 4; s++) { Gre[lane * 4 + s] = xr[s]; Gim[lane * 4 + s] = xi[s]; }
            }
        }
        INVM_PROF(3);
        lds_barrier();
        INVM_PROF(4);
        // ---- 3. pivoted pass: the owners of the pivot rows publish them as the B operand of the block step
        if (!STATIC) {
            const int s0 = rho[p0], s1 = rho[p0 + 1], s2 = rho[p0 + 2], s3 = rho[p0 + 3];
            #pragma unroll
            for (int r = 0; r < 4; r++) {
                const int x = 16 * w + kk + 4 * r;
                const int ps = (x == s0) ? 0 : (x == s1) ? 1 : (x == s2) ? 2 : (x == s3) ? 3 : -1;
                if (ps >= 0) {
                    #pragma unroll
                    for (int g = 0; g < NG; g++) Pr[ps * PW + 16 * g + c16] = M[g][r];
                }
            }
            lds_barrier();
        }
        // ---- 4. rank-4 block step on the MFMA, then the pivot columns take the multipliers
        {
            const int arow = 16 * w + c16;
            const int prow = STATIC ? p0 + kk : rho[p0 + kk];
            const double are = Gre[arow * 4 + kk] - ((arow == prow) ? 1.0 : 0.0);
            const double aim = Gim[arow * 4 + kk];
            #pragma unroll
            for (int g = 0; g < NG; g++) {
                double b1, b2;
                panel_b(Pr + kk * PW + 16 * g, c16, b1, b2);
                M[g] = MFMA(are, b1, M[g]);
                M[g] = MFMA(aim, b2, M[g]);
            }
            const int s = (c16 & 7) - q0;
            if (s >= 0 && s < 4) {
                const double *src = (c16 < 8) ? Gre : Gim;
                #pragma unroll
                for (int g = 0; g < NG; g++) {
                    if (g == gp) {
                        #pragma unroll
                        for (int r = 0; r < 4; r++) M[g][r] = src[(16 * w + kk + 4 * r) * 4 + s];
                    }
                }
            }
        }
        INVM_PROF(5);
        // no barrier here: the next panel writes F (last read before barrier 2) and the buffers read above
        // (G, Prow) alternate with the panel parity
    }
    return __any(bad);
}

// ---------------------------------------------------------------------------
// Inverse of a diagonal block (bs = NP <= 64 rows) of the N > 64 work matrix, for the block Gauss-Jordan inverse of
// qgd_k_dense.hip (qgdk_dense_inverse): the elimination of k_inverse_mfma (gj_panels: partial pivoting inside the block,
// rank-4 MFMA steps, matrix in registers) on a block read with the row stride of the big matrix; the result goes out as
// column-major planes D[i + 64 c] (+ 64*64: imaginary parts), the left operand of the block-row step.  A zero pivot marks the
// matrix in flags[] (it is then redone with full pivoting) instead of raising the status word.
// ---------------------------------------------------------------------------
template <int NP>
__global__ __launch_bounds__(NP * 4) __attribute__((amdgpu_waves_per_eu(3, 3)))
void k_inverse_diag(const double *__restrict__ Win, size_t mstride, int ldw, size_t off, double *__restrict__ DkC, int n0,
                    int *__restrict__ flags)
{
    constexpr int NG = NP / 8, NW = NP / 16, NTH = 64 * NW, PW = 2 * NP, LDP = NP + 1;
    constexpr int WORK = 8 * PW + 16 * NP + 8 * NP, SM = (WORK > NP * LDP) ? WORK : NP * LDP;
    __shared__ double smem[SM];
    __shared__ int rho[NP], rinv[NP];
    __shared__ int sing;
    const int n = n0 + blockIdx.x;
    const int t = threadIdx.x, w = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
    const int c16 = lane & 15, kk = lane >> 4;
    const double *Ln = Win + (size_t)n * mstride + off;
    d4 M[NG];
    #pragma unroll
    for (int g = 0; g < NG; g++)
        #pragma unroll
        for (int r = 0; r < 4; r++) M[g][r] = Ln[(size_t)(16 * w + kk + 4 * r) * ldw + 16 * g + c16];
    if (t == 0) sing = 0;
    __syncthreads();
    (void)gj_panels<NP, false>(INVM_PROF_ARG M, smem, rho, rinv, w, lane, (int)(blockIdx.x % NW), &sing);
    __syncthreads();
    if (t == 0 && sing) flags[n] = 1;
    // A^-1[rinv[x]][rho[j]] = M[x][j], one plane at a time through LDS (row-major staging), written column-major
    int orow[4];
    #pragma unroll
    for (int r = 0; r < 4; r++) orow[r] = rinv[16 * w + kk + 4 * r] * LDP;
    double *D = DkC + (size_t)n * 2 * 64 * 64;
    #pragma unroll
    for (int pass = 0; pass < 2; pass++) {
        if ((c16 >> 3) == pass) {
            #pragma unroll
            for (int g = 0; g < NG; g++) {
                const int oc = rho[8 * g + (c16 & 7)];
                #pragma unroll
                for (int r = 0; r < 4; r++) smem[orow[r] + oc] = M[g][r];
            }
        }
        lds_barrier();
        #pragma unroll
        for (int q = 0; q < NP * NP / NTH; q++) {
            const int e = t + q * NTH, c = e / NP, i = e % NP;
            D[pass * 64 * 64 + i + 64 * c] = smem[i * LDP + c];
        }
        lds_barrier();
    }
}

// ---------------------------------------------------------------------------
// K2 (64 < Np <= 288, default): the same blocked Gauss-Jordan with TWO block levels.  k_inverse_blocked streams the whole
// work slab (2 Np^2 doubles, 1 MB at Np = 256) through the accumulators once per 16 pivots: 6.4 GB per launch at config
// 5 against 0.6 GB algorithmic (PMC, profiles/r01_v8_pmc_c5.json), and it is HBM-bound there.  Block Gauss-Jordan has the
// same form for any block size -- M += (G - E_P) M[P,:], G = the in-place result of eliminating the pivot columns -- so
// the 16-pivot steps are applied to the 64 columns of a SUPER-PANEL only (that is the in-place elimination of an
// Np x 64 panel), and the columns outside it get one rank-64 update per super-panel: the slab is streamed Np/64 times
// instead of Np/16.  The rank-64 update keeps the A operand of a row block (16 rows x 64 pivots) in registers over all
// column groups; its B operand, the 64 pivot rows as they were before the super-step, is copied to a scratch area
// behind the slab first (the rows are overwritten in place by their own tiles).
// ---------------------------------------------------------------------------
#define INVB_NB 16
#define INVB_SB 64
// [-Bim | Bre] from [Bre | Bim]: rotate the 16-lane row by 8 and negate lanes 0..7 (as swap8_signed of qgd_k_dense.hip)
__device__ __forceinline__ double swap8_signed_inv(double b1, int sign_hi)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(b1), 0x128, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(b1), 0x128, 0xF, 0xF, false);
    return __hiloint2double(hi ^ sign_hi, lo);
}
__global__ __launch_bounds__(512) void k_inverse_blocked2(const double *__restrict__ L,
                                                          double *__restrict__ LinvA,
                                                          double *__restrict__ LinvT,
                                                          double *__restrict__ scratch, int Np, int n0,
                                                          int *__restrict__ status, const int *__restrict__ redo = nullptr)
{
    constexpr int NB = INVB_NB, SB = INVB_SB;
    if (redo) {                                      // fallback pass of the block Gauss-Jordan inverse: only the marked matrices
        if (!redo[n0 + blockIdx.x]) return;
        if (threadIdx.x == 0) atomicAdd(status + 1, 1);      // (counted: qgd_get_intermediate "repivoted")
    }
    extern __shared__ double smem[];
    const int PW = 2 * Np;
    double *Fre = smem, *Fim = Fre + (size_t)Np * NB;          // panel columns -> multipliers
    double *Bp = Fim + (size_t)Np * NB;                         // [NB][2*SB] pivot rows of the inner step (super-panel columns only)
    double *fre = Bp + (size_t)NB * 2 * SB, *fim = fre + Np;    // column s of the panel before the step
    double *yrow = fim + Np;                                    // [2*NB] scaled pivot row
    double *redv = yrow + 2 * NB;                               // [8]
    int *redi = reinterpret_cast<int *>(redv + 8);              // [8]
    int *rho = redi + 8, *rinv = rho + Np, *used = rinv + Np;
    const int n = n0 + blockIdx.x;
    const int t = threadIdx.x, nth = blockDim.x, wave = t >> 6, lane = t & 63, nw = nth >> 6;
    const int c16 = lane & 15, kk = lane >> 4;
    const size_t panel = (size_t)Np * PW, pl = (size_t)Np * Np;
    double *W = scratch + (size_t)blockIdx.x * (panel + (size_t)SB * PW);
    double *Bg = W + panel;                                      // [SB][PW] pivot rows of the super-step (old values)
    const double *Ln = L + (size_t)n * panel;
    // (no copy of L into the slab: every element is READ from L until the step that first writes it -- the columns of
    //  super-panel 0 until its first inner update, the other columns until the first rank-64 update -- and from the slab after)
    for (int r = t; r < Np; r += nth) used[r] = 0;
    __syncthreads();

    for (int q0 = 0; q0 < Np; q0 += SB) {
        const int sbw = (Np - q0 < SB) ? Np - q0 : SB;          // pivots of this super-panel (a multiple of 16)
        const int g_lo = q0 >> 3, sg = sbw >> 3;                // its column groups: g_lo .. g_lo + sg - 1
        for (int p0 = q0; p0 < q0 + sbw; p0 += NB) {
            const double *Ws = (p0 == 0) ? Ln : W;              // source of the super-panel's columns in this step
            // ---- 1. the 16 panel columns into REGISTERS: thread t keeps the elements (r, q) = (t/16 + 32 k, t%16), k < KMAX,
            //         for the 16 pivot steps of the panel (k_inverse_blocked keeps them in LDS: 64 LDS operations per thread
            //         and pivot step, three barriers; here ~20 and two)
            constexpr int KMAX = 9;                              // Np <= 288
            const int q = t & 15, r0 = t >> 4;
            double xr[KMAX], xi[KMAX];
            #pragma unroll
            for (int k = 0; k < KMAX; k++) {
                const int r = r0 + 32 * k, col = p0 + q;
                const bool ok = r < Np;
                const size_t o = (size_t)(ok ? r : 0) * PW + (col >> 3) * 16 + (col & 7);
                xr[k] = ok ? Ws[o] : 0.0; xi[k] = ok ? Ws[o + 8] : 0.0;
            }
            // ---- 2. pivoted in-place Gauss-Jordan on the Np x 16 panel (rows stay where they are)
            for (int s = 0; s < NB; s++) {
                if (q == s) {                                    // column s of the panel, for everybody
                    #pragma unroll
                    for (int k = 0; k < KMAX; k++) { const int r = r0 + 32 * k; if (r < Np) { fre[r] = xr[k]; fim[r] = xi[k]; } }
                }
                __syncthreads();
                // pivot = largest modulus among the unused rows, found by every wave for itself (no cross-wave exchange);
                // ties go to the lowest row so that all waves agree
                // (one 32-bit key per row: the upper bits of |x|^2 with the low 9 bits replaced by 511 - row, so that a DPP
                //  max-scan yields the arg-max -- pivot = largest modulus compared on 23 bits, ties to the lowest row)
                unsigned key = 0;
                for (int r = lane; r < Np; r += 64) {
                    const double a = fre[r], b = fim[r];
                    const unsigned mag = (unsigned)__double2hiint(a * a + b * b);
                    const unsigned kr = used[r] ? 0u : ((mag & ~511u) | (unsigned)(511 - r));
                    key = kr > key ? kr : key;
                }
                key = wave_max_u32(key);
                const int pr = 511 - (int)(key & 511u);
                const bool nonsingular = (key >> 9) != 0;
                double f1[KMAX], f2[KMAX];                       // column s at my rows: the multipliers of this step
                #pragma unroll
                for (int k = 0; k < KMAX; k++) { const int r = r0 + 32 * k; f1[k] = (r < Np) ? fre[r] : 0.0; f2[k] = (r < Np) ? fim[r] : 0.0; }
                {                                                // the owners of row pr publish it scaled (the pivot entry becomes 1/pivot)
                    const double a = fre[pr], b = fim[pr];
                    const double den = fast_rcp(a * a + b * b), ir = a * den, ii = -b * den;
                    #pragma unroll
                    for (int k = 0; k < KMAX; k++) {
                        if (r0 + 32 * k == pr) {
                            yrow[q] = (q == s) ? ir : xr[k] * ir - xi[k] * ii;
                            yrow[NB + q] = (q == s) ? ii : xr[k] * ii + xi[k] * ir;
                        }
                    }
                }
                if (t == 0) { rho[p0 + s] = pr; rinv[pr] = p0 + s; if (!nonsingular) *status = 1; }
                __syncthreads();
                if (t == 0) used[pr] = 1;                        // (read again only after the next step's barrier)
                const double rr = yrow[q], ri = yrow[NB + q];
                #pragma unroll
                for (int k = 0; k < KMAX; k++) {
                    if (r0 + 32 * k == pr) { xr[k] = rr; xi[k] = ri; }
                    else {
                        const double br = (q == s) ? 0.0 : xr[k], bi2 = (q == s) ? 0.0 : xi[k];
                        xr[k] = br - (f1[k] * rr - f2[k] * ri);
                        xi[k] = bi2 - (f1[k] * ri + f2[k] * rr);
                    }
                }
            }
            // the factored panel (multipliers / inverse entries) into LDS: the A operand of the updates
            #pragma unroll
            for (int k = 0; k < KMAX; k++) {
                const int r = r0 + 32 * k;
                if (r < Np) { Fre[r * NB + q] = xr[k]; Fim[r * NB + q] = xi[k]; }
            }
            __syncthreads();
            // ---- 3. the 16 pivot rows (their values before the step), super-panel columns only, as B operand
            for (int e = t; e < NB * 2 * sbw; e += nth) {
                const int k = e / (2 * sbw), c = e % (2 * sbw);
                Bp[k * 2 * SB + c] = Ws[(size_t)rho[p0 + k] * PW + 16 * g_lo + c];
            }
            __syncthreads();
            // ---- 4. rank-16 step on the columns of the super-panel; the pivot columns then take the multipliers
            const int gp = p0 >> 3;
            // (four column groups of a row block at a time: independent accumulators, one A operand, 16 loads in flight)
            const int sq = (sg + 3) >> 2;
            for (int ti = wave; ti < (Np / 16) * sq; ti += nw) {
                const int rb = ti / sq, gq = (ti % sq) * 4;
                const int arow = 16 * rb + c16;
                d4 acc[4];
                int gi[4]; bool ok[4];
                #pragma unroll
                for (int q = 0; q < 4; q++) {
                    ok[q] = gq + q < sg;
                    gi[q] = ok[q] ? gq + q : gq;
                    #pragma unroll
                    for (int r = 0; r < 4; r++) acc[q][r] = Ws[(size_t)(16 * rb + kk + 4 * r) * PW + 16 * (g_lo + gi[q]) + c16];
                }
                #pragma unroll
                for (int ks = 0; ks < NB / 4; ks++) {
                    const int sidx = 4 * ks + kk;
                    const double are = Fre[arow * NB + sidx] - ((arow == rho[p0 + sidx]) ? 1.0 : 0.0);
                    const double aim = Fim[arow * NB + sidx];
                    #pragma unroll
                    for (int q = 0; q < 4; q++) {
                        double b1, b2;
                        panel_b(Bp + (size_t)sidx * 2 * SB + 16 * gi[q], c16, b1, b2);
                        acc[q] = MFMA(are, b1, acc[q]);
                        acc[q] = MFMA(aim, b2, acc[q]);
                    }
                }
                #pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int g = g_lo + gi[q];
                    if (g == gp || g == gp + 1) {
                        const int sidx = 8 * (g - gp) + (c16 & 7);
                        #pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const int row = 16 * rb + kk + 4 * r;
                            acc[q][r] = (c16 < 8) ? Fre[row * NB + sidx] : Fim[row * NB + sidx];
                        }
                    }
                    if (ok[q]) {
                        #pragma unroll
                        for (int r = 0; r < 4; r++) W[(size_t)(16 * rb + kk + 4 * r) * PW + 16 * g + c16] = acc[q][r];
                    }
                }
            }
            __syncthreads();
        }
        if (sbw == Np) break;                                   // one super-panel covers the matrix: nothing outside it
        const double *Wo = (q0 == 0) ? Ln : W;                  // the outside columns have not been written before the first rank-64 update
        // ---- 5. the sbw pivot rows of the super-step as they are now in the OUTSIDE columns (untouched so far)
        #pragma unroll 8
        for (int e = t; e < sbw * PW; e += nth) {
            const int k = e / PW, c = e % PW;
            if ((c >> 4) < g_lo || (c >> 4) >= g_lo + sg) Bg[(size_t)k * PW + c] = Wo[(size_t)rho[q0 + k] * PW + c];
        }
        __syncthreads();
        // ---- 6. rank-sbw update of the outside columns: M += (G - E_P) M_old[P,:], G = the super-panel's columns
        const int nks = sbw >> 2;
        for (int rb = wave; rb < Np / 16; rb += nw) {
            const int arow = 16 * rb + c16;
            double are[SB / 4], aim[SB / 4];
            #pragma unroll
            for (int ks = 0; ks < SB / 4; ks++) {
                if (ks < nks) {
                    const int sidx = 4 * ks + kk, col = q0 + sidx;
                    const size_t o = (size_t)arow * PW + (col >> 3) * 16 + (col & 7);
                    are[ks] = W[o] - ((arow == rho[q0 + sidx]) ? 1.0 : 0.0);
                    aim[ks] = W[o + 8];
                } else { are[ks] = 0.0; aim[ks] = 0.0; }
            }
            // four outside column groups at a time (independent accumulators); the B operand of step ks+2 is in flight
            // while the 8 MFMAs of step ks issue; [-Bim | Bre] by a DPP row rotation instead of a second load
            const int nout = Np / 8 - sg, sign_hi = (c16 < 8) ? (int)0x80000000 : 0;
            for (int o0 = 0; o0 < nout; o0 += 4) {
                int go[4]; bool ok[4];
                d4 acc[4];
                const double *bp[4];
                #pragma unroll
                for (int q = 0; q < 4; q++) {
                    ok[q] = o0 + q < nout;
                    const int o = ok[q] ? o0 + q : o0;
                    go[q] = (o < g_lo) ? o : o + sg;
                    bp[q] = Bg + (size_t)kk * PW + 16 * go[q] + c16;
                    #pragma unroll
                    for (int r = 0; r < 4; r++) acc[q][r] = Wo[(size_t)(16 * rb + kk + 4 * r) * PW + 16 * go[q] + c16];
                }
                double b0[4], b1[4], b2[4];
                #pragma unroll
                for (int q = 0; q < 4; q++) { b0[q] = bp[q][0]; b1[q] = bp[q][(size_t)((nks > 1) ? 4 : 0) * PW]; }
                #pragma unroll
                for (int ks = 0; ks < SB / 4; ks++) {
                    if (ks < nks) {
                        const int kn = (ks + 2 < nks) ? ks + 2 : nks - 1;
                        #pragma unroll
                        for (int q = 0; q < 4; q++) b2[q] = bp[q][(size_t)kn * 4 * PW];
                        #pragma unroll
                        for (int q = 0; q < 4; q++) {
                            const double bs = swap8_signed_inv(b0[q], sign_hi);
                            acc[q] = MFMA(are[ks], b0[q], acc[q]);
                            acc[q] = MFMA(aim[ks], bs, acc[q]);
                        }
                        #pragma unroll
                        for (int q = 0; q < 4; q++) { b0[q] = b1[q]; b1[q] = b2[q]; }
                    }
                }
                #pragma unroll
                for (int q = 0; q < 4; q++) {
                    if (ok[q]) {
                        #pragma unroll
                        for (int r = 0; r < 4; r++) W[(size_t)(16 * rb + kk + 4 * r) * PW + 16 * go[q] + c16] = acc[q][r];
                    }
                }
            }
        }
        __syncthreads();
    }
    // A^-1[i][c] = M[rho[i]][rinv[c]].  Tiles of 32 x 64 go through LDS: the slab is gathered (rows rho[i], columns
    // rinv[c]: scattered inside a row, L1-friendly) and BOTH outputs are written in runs of consecutive doubles --
    // LinvT row-major, LinvA column-major.  (Scattering the slab straight into the two outputs, as k_inverse_blocked does,
    // writes LinvA 8 bytes per 64-byte line: 0.85 M of the kernel's 5 M cycles at Np = 256.)
    double *A = LinvA + (size_t)n * 2 * pl, *T = LinvT + (size_t)n * 2 * pl;
    double *tre = smem, *tim = smem + 32 * 65;
    for (int i0 = 0; i0 < Np; i0 += 32)
        for (int c0 = 0; c0 < Np; c0 += 64) {
            #pragma unroll
            for (int k = 0; k < 4; k++) {
                const int idx = t + k * 512, il = idx >> 6, cl = idx & 63;
                const int i = i0 + il, c = c0 + cl;
                if (i < Np && c < Np) {
                    const int jj = rinv[c];
                    const size_t o = (size_t)rho[i] * PW + (jj >> 3) * 16 + (jj & 7);
                    tre[il * 65 + cl] = W[o]; tim[il * 65 + cl] = W[o + 8];
                }
            }
            __syncthreads();
            #pragma unroll
            for (int k = 0; k < 4; k++) {
                const int idx = t + k * 512;
                {
                    const int il = idx >> 6, cl = idx & 63, i = i0 + il, c = c0 + cl;
                    if (i < Np && c < Np) { T[(size_t)i * Np + c] = tre[il * 65 + cl]; T[pl + (size_t)i * Np + c] = tim[il * 65 + cl]; }
                }
                {
                    const int cl = idx >> 5, il = idx & 31, i = i0 + il, c = c0 + cl;
                    if (i < Np && c < Np) { A[(size_t)i + (size_t)Np * c] = tre[il * 65 + cl]; A[pl + (size_t)i + (size_t)Np * c] = tim[il * 65 + cl]; }
                }
            }
            __syncthreads();
        }
}

static inline size_t inverse_blocked2_lds(int Np)
{
    return ((size_t)2 * Np * INVB_NB + (size_t)INVB_NB * 2 * INVB_SB + 2 * Np + 2 * INVB_NB + 8) * sizeof(double) + (size_t)(8 + 3 * Np) * sizeof(int);
}

static inline size_t inverse_blocked_lds(int Np)
{
    return ((size_t)2 * Np * INVB_NB + (size_t)INVB_NB * 2 * Np + 2 * Np + 2 * INVB_NB + 8) * sizeof(double) + (size_t)(8 + 3 * Np) * sizeof(int);
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

template <int M, int NOPS>
static int launch_front(const qgdk_ctx *c)
{
    const size_t shm = front_lds(M, c->ell_z);
    const int nt = c->nt;
    FrontPre pre{0, 0, 0};
    pre.extra = qgdk_front_pre_plan(c, &pre.q2, &pre.q1);
    if (nt > CB_ONE_ALONE && nt <= CB_ONE_ROUND) {
        SET_LDS_ONCE((k_front<M, NOPS, true>), shm);
        hipLaunchKernelGGL((k_front<M, NOPS, true>), dim3(nt), dim3(256), shm, c->stream, c->ell_col, c->ell_inv, c->ell_val, c->tab, c->cw, c->n_ops, c->ell_z,
                           c->L, c->R, c->LinvT, c->Pr, c->Pc, c->status, c->status + 1, pre);
    } else {
        SET_LDS_ONCE((k_front<M, NOPS, false>), shm);
        hipLaunchKernelGGL((k_front<M, NOPS, false>), dim3(nt), dim3(256), shm, c->stream, c->ell_col, c->ell_inv, c->ell_val, c->tab, c->cw, c->n_ops, c->ell_z,
                           c->L, c->R, c->LinvT, c->Pr, c->Pc, c->status, c->status + 1, pre);
    }
    return (int)hipGetLastError();
}

extern "C" {

// Fused front: which problems take it (the host asks before it plans an evaluation), and its launch
int qgdk_front_supported(const qgdk_ctx *c)
{
    if (c->Np != 64 || !c->use_sparse || c->m < 1 || c->m > 4 || c->ell_z < 1 || c->ell_z > 16) return 0;
    return front_lds(c->m, c->ell_z) <= 53 * 1024 ? 1 : 0;      // three workgroups per CU
}

// Which workgroups of k_front start from step matrices the tables launch built (qgd_front.h: FrontPre): all three workgroups
// of every CU that holds three (the third and the second where that is more than 192) -- as long as they fit, one per CU,
// beside 64 table workgroups.  (More was
// measured, up to every second and third workgroup and the first ones of the CUs that hold three -- scripts/front_pre_sweep.py,
// EXPERIMENTS.md "Round 6": the tables launch grows by what the front kernel gains.)  QGD_PATHS=front_nopre: none (A/B).
int qgdk_front_pre_plan(const qgdk_ctx *c, int *q2, int *q1)
{
    *q2 = 0; *q1 = 0;
    if (qgd_path("front_nopre")) return 0;
    const int nt = c->nt, extra = front_extra(nt);
    if (extra > 192) return 0;
    if (extra > 0) {
        *q2 = (2 * extra > 192) ? 192 - extra : extra;
        if (3 * extra <= 192) *q1 = extra;      // (all three workgroups of the fullest CUs: 293.3 against 297.2-298.3 us per evaluation with two of three)
        return extra;
    }
    // (256 < nt <= 512 -- the CUs 0 .. nt-257 hold two workgroups, the rest one -- with the second workgroups pre-built: measured,
    //  3-5 us slower than without at 281 .. 513 time points, and the general path is faster there either way)
    return 0;
}

int qgdk_front(const qgdk_ctx *c)
{
    if (c->m == 4 && c->n_ops == 3) return launch_front<4, 3>(c);
    switch (c->m) {
    case 1: return launch_front<1, -1>(c);
    case 2: return launch_front<2, -1>(c);
    case 3: return launch_front<3, -1>(c);
    case 4: return launch_front<4, -1>(c);
    default: return (int)hipErrorInvalidValue;
    }
}

int qgdk_inverse(const qgdk_ctx *c)
{
    const int nmat = c->nt - 1;
    if (nmat <= 0) return 0;
    switch (c->Np) {     // Np <= 64: inverse and propagator in one launch, the matrix in registers
#define CALL_IMF(N) case N: hipLaunchKernelGGL((k_inverse_mfma<N>), dim3(nmat), dim3(N * 4), 0, c->stream, c->L, c->R, c->LinvT, c->Pr, c->Pc, 1, c->status); \
                            return (int)hipGetLastError()
        CALL_IMF(16); CALL_IMF(32); CALL_IMF(48);
#undef CALL_IMF
    case 64:   // column-block elimination of [L | R] (qgd_inverse_cb.h); QGD_PATHS=inv_panels: its last resort, the 4-pivot panel kernel, for every matrix
        if (qgd_path("inv_panels")) hipLaunchKernelGGL((k_inverse_mfma<64>), dim3(nmat), dim3(256), 0, c->stream, c->L, c->R, c->LinvT, c->Pr, c->Pc, 1, c->status);
        else if (nmat > CB_ONE_ALONE && nmat <= CB_ONE_ROUND) hipLaunchKernelGGL(k_inverse_cb<true>, dim3(nmat), dim3(256), 0, c->stream, c->L, c->R, c->LinvT, c->Pr, c->Pc, 1, c->status, c->status + 1);
        else hipLaunchKernelGGL(k_inverse_cb<false>, dim3(nmat), dim3(256), 0, c->stream, c->L, c->R, c->LinvT, c->Pr, c->Pc, 1, c->status, c->status + 1);
        return (int)hipGetLastError();
    default: break;
    }
    if (c->Np > 64 && c->dense_gemm) {    // (any size: no LDS limit)
        const int took = qgdk_dense_inverse(c);              // block Gauss-Jordan as batched GEMM launches (qgd_k_dense.hip)
        if (took) return took < 0 ? (int)hipErrorUnknown : (int)hipGetLastError();
    }
    if (c->inv_scratch && inverse_blocked_lds(c->Np) <= 150 * 1024) {
        const size_t shm = inverse_blocked2_lds(c->Np);      // two block levels: 64-column super-panels
        HIPCHK(hipFuncSetAttribute((const void *)k_inverse_blocked2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
        for (int n0 = 1; n0 < c->nt; n0 += c->inv_batch) {
            const int nb = (c->nt - n0 < c->inv_batch) ? c->nt - n0 : c->inv_batch;
            hipLaunchKernelGGL(k_inverse_blocked2, dim3(nb), dim3(512), shm, c->stream, c->L, c->LinvA, c->LinvT, c->inv_scratch,
                               c->Np, n0, c->status, (const int *)nullptr);
        }
        return (int)hipGetLastError();
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

int qgdk_inverse_diag(const qgdk_ctx *c, const double *Win, size_t mstride, int ldw, size_t off, int bs, double *DkC, int *flags)
{
    const int nmat = c->nt - 1;
    switch (bs) {
#define CALL_ID(N) case N: hipLaunchKernelGGL((k_inverse_diag<N>), dim3(nmat), dim3(N * 4), 0, c->stream, Win, mstride, ldw, off, DkC, 1, flags); break
        CALL_ID(16); CALL_ID(32); CALL_ID(48); CALL_ID(64);
#undef CALL_ID
    default: return -1;
    }
    return (int)hipGetLastError();
}

// the matrices marked in flags[] again, by k_inverse_blocked2 (partial pivoting over whole columns) from the untouched L
int qgdk_inverse_redo(const qgdk_ctx *c, const int *flags)
{
    if (inverse_blocked_lds(c->Np) > 150 * 1024) {       // N > ~280: the generic kernel (matrix in a global slab) is the pivoting fallback
        const size_t shm = (size_t)(3 * c->Np + 16) * sizeof(double);
        for (int n0 = 1; n0 < c->nt; n0 += c->inv_batch) {
            const int nb = (c->nt - n0 < c->inv_batch) ? c->nt - n0 : c->inv_batch;
            hipLaunchKernelGGL(k_inverse, dim3(nb), dim3(256), shm, c->stream, c->L, c->LinvA, c->LinvT, c->inv_scratch, c->Np, n0, 0,
                               c->status, flags);
        }
        return (int)hipGetLastError();
    }
    const size_t shm = inverse_blocked2_lds(c->Np);
    HIPCHK(hipFuncSetAttribute((const void *)k_inverse_blocked2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    for (int n0 = 1; n0 < c->nt; n0 += c->inv_batch) {
        const int nb = (c->nt - n0 < c->inv_batch) ? c->nt - n0 : c->inv_batch;
        hipLaunchKernelGGL(k_inverse_blocked2, dim3(nb), dim3(512), shm, c->stream, c->L, c->LinvA, c->LinvT, c->inv_scratch, c->Np, n0,
                           c->status, flags);
    }
    return (int)hipGetLastError();
}

int qgdk_propagator_is_fused(const qgdk_ctx *c) { return c->Np == 16 || c->Np == 32 || c->Np == 48 || c->Np == 64; }

int qgdk_propagator(const qgdk_ctx *c)
{
    if (qgdk_propagator_is_fused(c)) return 0;                    // k_inverse_mfma produced P_n already
    if (c->Np > 64 && c->dense_gemm && qgdk_dense_propagator(c)) return (int)hipGetLastError();      // three-product tiles (qgd_k_dense.hip)
    const int ngroups = c->Np / 8;
    const int gtiles = (ngroups + LV_NG - 1) / LV_NG;
    const int rtiles = (c->Np + 63) / 64;
    hipLaunchKernelGGL(k_propagator, dim3(gtiles * rtiles, c->nt - 1), dim3(256), 0, c->stream, c->LinvA, c->R,
                       c->Pr, c->Pc, c->Np);
    return (int)hipGetLastError();
}


} // extern "C"
