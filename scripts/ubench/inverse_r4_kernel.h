// inverse_r4_kernel.h -- the round-4 Np <= 64 inverse + propagator kernel (4-pivot panels, every wave in every panel,
// product phase) WITH its timing instrumentation and knock-out switches, frozen here for scripts/ubench/inverse_bench.hip.
// The knock-outs (-DQGD_INV_KO_CHAIN, -DQGD_INV_KO_PRODUCT, -DQGD_INV_KO_PW_MFMA, -DQGD_INV_KO_QUARTER) skip work: the
// results are WRONG, only the time is read (DESIGN.md section 7 has the numbers).  The product library carries none of
// them: csrc/qgd_k_inverse.hip holds the same elimination without switches, as the fallback of k_inverse_cb.
#pragma once
#include "../../quantumgatedesign.jl_amd/csrc/qgd_kernels_common.h"
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
__device__ __forceinline__ double r4_lane_read(double v, int l)     // l wave-uniform
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

template <int CTRL, int ROWMASK>
__device__ __forceinline__ unsigned r4_dpp_max_step(unsigned key)
{
    const unsigned o = (unsigned)__builtin_amdgcn_update_dpp((int)key, (int)key, CTRL, ROWMASK, 0xF, false);
    return o > key ? o : key;
}

// maximum of a 32-bit key over the wave, returned wave-uniform
__device__ __forceinline__ unsigned r4_wave_max_u32(unsigned key)
{
    key = r4_dpp_max_step<0x111, 0xF>(key);      // row_shr:1
    key = r4_dpp_max_step<0x112, 0xF>(key);      // row_shr:2
    key = r4_dpp_max_step<0x114, 0xF>(key);      // row_shr:4
    key = r4_dpp_max_step<0x118, 0xF>(key);      // row_shr:8   -> lane 15 of each row holds the row maximum
    key = r4_dpp_max_step<0x142, 0xA>(key);      // row_bcast:15 into rows 1 and 3
    key = r4_dpp_max_step<0x143, 0xC>(key);      // row_bcast:31 into rows 2 and 3
    return (unsigned)__builtin_amdgcn_readlane((int)key, 63);
}

__device__ __forceinline__ double r4_fast_rcp(double d)             // v_rcp_f64 + 2 Newton steps (full precision)
{
    double r = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, r, 1.0); r = __builtin_fma(r, e, r);
    e = __builtin_fma(-d, r, 1.0); r = __builtin_fma(r, e, r);
    return r;
}

template <int NP>
__global__ __launch_bounds__(NP * 4) __attribute__((amdgpu_waves_per_eu(3, 3)))
void k_inverse_r4(const double *__restrict__ L, const double *__restrict__ R, double *__restrict__ LinvT,
                    double *__restrict__ Pr, double *__restrict__ Pc, int n0, int *__restrict__ status
#ifdef QGD_INV_PROFILE      // scripts/ubench/inverse_bench.hip: cycles of workgroup 0 / wave 0 per phase
                    , unsigned long long *prof)
{
    long long prof_last = clock64();
#define INV_PROF(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) { const long long now_ = clock64(); atomicAdd(&prof[i], (unsigned long long)(now_ - prof_last)); prof_last = now_; } } while (0)
#else
                    )
{
#define INV_PROF(i) do { } while (0)
#endif
    constexpr int NG = NP / 8, NW = NP / 16, NTH = 64 * NW, PW = 2 * NP, LDP = NP + 1;
    // LDS: the working buffers of the elimination, overlaid by the output staging plane at the end
    constexpr int O_PROW = 0, O_G = O_PROW + 8 * PW, O_F = O_G + 16 * NP,
                  WORK = O_F + 8 * NP, SM = (WORK > NP * LDP) ? WORK : NP * LDP;
    __shared__ double smem[SM];
    __shared__ int rho[NP], rinv[NP];                   // rho[p] = row of the p-th pivot
    double *Prow = smem + O_PROW;                       // [2][4][PW]   pivot rows (B operand), by panel parity
    double *Gm = smem + O_G;                            // [2][2][NP][4] multipliers re/im, by panel parity
    double *Fm = smem + O_F;                            // [2][NP][4]   panel columns re/im
    const int n = n0 + blockIdx.x;
    const int t = threadIdx.x, w = t >> 6, lane = t & 63;
    const int c16 = lane & 15, kk = lane >> 4;
    const size_t panel = (size_t)NP * PW, pl = (size_t)NP * NP;
    const double *Ln = L + (size_t)n * panel;

    d4 M[NG];
    #pragma unroll
    for (int g = 0; g < NG; g++)
        #pragma unroll
        for (int r = 0; r < 4; r++) M[g][r] = Ln[(size_t)(16 * w + kk + 4 * r) * PW + 16 * g + c16];
    INV_PROF(0);
    bool used = lane >= NP;                             // panel wave: this lane's row has been a pivot row
    // which wave factors the panels (rotating it differently across workgroups that share a CU, e.g. with
    // blockIdx/256, changes nothing: 108-111 us for 550 matrices either way)
#ifdef QGD_INV_PW_FIXED          // (experiment: the panel wave has the same index in every workgroup)
    const int pw = QGD_INV_PW_FIXED;
#else
    const int pw = (NW > 1) ? (int)(blockIdx.x % NW) : 0;
#endif

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
        INV_PROF(1);
        // ---- 2. one wave (a different one in neighbouring workgroups, so that the serial phases of
        //         the workgroups sharing a CU sit on different SIMDs): pivoted in-place Gauss-Jordan
        //         on the NP x 4 panel, lane = row
        if (w == pw) {
            __builtin_amdgcn_s_setprio(1);      // the pivot chain is the critical path: ahead of other workgroups' MFMA bursts on this SIMD
            double xr[4], xi[4];
            const int lrow = (lane < NP) ? lane : 0;
            #pragma unroll
            for (int s = 0; s < 4; s++) { xr[s] = Fm[lrow * 4 + s]; xi[s] = Fm[4 * NP + lrow * 4 + s]; }
#ifdef QGD_INV_KO_CHAIN
            if (lane == 0) { for (int s = 0; s < 4; s++) { rho[p0 + s] = p0 + s; rinv[p0 + s] = p0 + s; } }
#else
            #pragma unroll
            for (int s = 0; s < 4; s++) {
                const double m2 = xr[s] * xr[s] + xi[s] * xi[s];
                const unsigned mag = (unsigned)__double2hiint(m2);
                unsigned key = used ? 0u : ((mag & ~63u) | (unsigned)(63 - lane));
#ifdef QGD_INV_RCP_HOIST
                // (round 4 experiment, off: every lane inverts its OWN candidate while the arg-max runs, so that the reciprocal
                //  and its two Newton steps leave the serial pivot chain; same bits.  Measured with
                //  scripts/ubench/inverse_bench.hip: 66.6 / 91.3 / 109.2 us for 256 / 512 / 550 matrices against 65.7 / 91.1 /
                //  109.0 -- the chain is not waiting on that dependency; not kept.)
                const double den_own = r4_fast_rcp(m2);
                double yr[4], yi[4];
                yr[s] = xr[s] * den_own; yi[s] = -xi[s] * den_own;
#endif
                key = r4_wave_max_u32(key);
                const int pr = 63 - (int)(key & 63u);
                if (lane == 0) { rho[p0 + s] = pr; rinv[pr] = p0 + s; if ((key >> 6) == 0) *status = 1; }
                used = used || (lane == pr);
#ifdef QGD_INV_RCP_HOIST
                #pragma unroll
                for (int q = 0; q < 4; q++) {
                    if (q == s) { yr[q] = r4_lane_read(yr[q], pr); yi[q] = r4_lane_read(yi[q], pr); }
                    else { yr[q] = r4_lane_read(xr[q], pr); yi[q] = r4_lane_read(xi[q], pr); }
                }
                const double ir = yr[s], ii = yi[s];
#else
                double yr[4], yi[4];
                #pragma unroll
                for (int q = 0; q < 4; q++) { yr[q] = r4_lane_read(xr[q], pr); yi[q] = r4_lane_read(xi[q], pr); }
                const double den = r4_fast_rcp(yr[s] * yr[s] + yi[s] * yi[s]);
                const double ir = yr[s] * den, ii = -yi[s] * den;
#endif
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
#endif
            if (lane < NP) {
                #pragma unroll
                for (int s = 0; s < 4; s++) { Gre[lane * 4 + s] = xr[s]; Gim[lane * 4 + s] = xi[s]; }
            }
            __builtin_amdgcn_s_setprio(0);
        }
        __syncthreads();
        INV_PROF(2);
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
        INV_PROF(3);
        // ---- 4. rank-4 block step on the MFMA, then the pivot columns take the multipliers
        {
            const int arow = 16 * w + c16;
            const double are = Gre[arow * 4 + kk] - ((arow == rho[p0 + kk]) ? 1.0 : 0.0);
            const double aim = Gim[arow * 4 + kk];
#ifdef QGD_INV_KO_PW_MFMA         // (timing experiment, wrong results: the panel wave issues no MFMA)
            if (w != pw)
#endif
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
        INV_PROF(4);
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
#ifndef QGD_INV_KO_PRODUCT      // (knock-out experiments of scripts/ubench/inverse_bench.hip: timing only, wrong results)
#ifdef QGD_INV_KO_PW_MFMA
        if (w != pw)
#endif
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
#ifdef QGD_INV_KO_QUARTER      // (timing experiment, wrong results: a quarter of the product's MFMAs dropped -- what a three-product form would save)
                if (pass == 1 && rt >= NRT / 2) continue;
#endif
                const double af = smem[(16 * rt + c16) * LDP + k];
                #pragma unroll
                for (int gg = 0; gg < GPW; gg++) acc[rt][gg] = MFMA(af, bf[gg], acc[rt][gg]);
            }
        }
#endif
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
    INV_PROF(5);
#undef INV_PROF
}
