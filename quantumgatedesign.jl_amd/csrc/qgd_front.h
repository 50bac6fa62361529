// qgd_front.h -- Np = 64, sparse operators: everything an evaluation needs from ONE time point, in one workgroup
// (included by qgd_k_inverse.hip behind qgd_inverse_cb.h; replaces the build + solve of forward_evolution.jl:181-220 and
// :421-461 for all right-hand sides at once, with L and R taken at the SAME time point).
//
// The step  L(t_{n+1}) w_{n+1} = R(t_n) w_n  pairs two time points as long as the sweeps run in w.  In the variable
// phi_n = L_n psi_n the forward sweep is  phi_{n+1} = S_n phi_n  with the same-point product  S_n = R_n L_n^-1, and the adjoint
// sweep runs directly in lambda:  lambda_n = S_n^H lambda_{n+1} + L_n^-H f_n  (tests/proto_propagator.py::evaluate_local).
// One workgroup can then
//   A. build L_n and R_n (k_build_LR_ell's recursion: lane = row, four columns per thread, the 64 columns as four slabs of
//      16, A_d(t_n) assembled into LDS ONCE per time point instead of once per 32-column half), and leave them to itself as
//      E = L_n^H and F = R_n^H in panel layout -- a thread's column c of L is row c of L^H with lane = column, so the
//      transposition is in the store addresses and the stores are runs of 64 bytes; no staging through LDS;
//   B. eliminate [E | F] -> [E^-1 | E^-1 F] = [L_n^-H | S_n^H] with the column-block Gauss-Jordan of qgd_inverse_cb.h as it
//      stands, pivot stages and all (its loads find E and F in the L2 of the XCD that has just written them).
// What leaves the kernel: X = L_n^-H as row-major planes in LinvT (left operand of psi_n = L_n^-1 phi_n = X^H phi_n and of
// h_n = X f_n: k_psi), and S_n exactly where and how the two-point form keeps P_n -- column-major planes in Pc (left operand of
// the forward scan), row-major panel in Pr (left operand S_n^H of the adjoint scan) -- so the scan kernels did not change.
// With Y = S_n^H in the accumulators, S in column-major planes is conj(Y) in row-major planes (straight from the registers, the
// sign of the imaginary plane flipped) and S as a row-major panel is conj(Y) as a column-major panel (through the wave's LDS
// slab): qgd_inverse_cb.h, cb_wave<.., FRONT>.  No launch boundary between build and elimination, and the build's
// latency-bound phases run beside the MFMA stretches of the CU's other workgroups; the panels of L^H and R^H still pass
// through memory (three workgroups per CU leave no room for 128 KB).  Measured: EXPERIMENTS.md "Round 6".
#pragma once
#include "qgd_ell.h"

// LDS of phase A in bytes: A_d(t_n) [M][Z][64] complex, the source slab(s) [64][17] complex, the neighbour lists [Z][64]
static inline size_t front_build_lds(int M, int Z, int slabs = 1) { return ((size_t)M * Z * 64 + (size_t)slabs * 64 * 17) * 16 + (size_t)Z * 64 * 4; }

// Phase A.  Eh, Fh: this time point's [64][128] panels of L^H and R^H.  NTH = 256 threads: the four 16-column slabs one after
// the other (k_front: three workgroups per CU hide each other's latencies); NTH = 1024: side by side, one per group of four
// waves (k_tables_front: a workgroup alone on its CU -- the four slabs in sequence took 40 us there).
template <int M, int NOPS, int NTH = 256>
__device__ __forceinline__ void front_build(double *smem_raw, const int32_t *__restrict__ ell_col, const uint8_t *__restrict__ ell_inv,
                                            const double *__restrict__ ell_val, const double *__restrict__ tab,
                                            const double *__restrict__ cw, const int n, const int n_ops, const int Z,
                                            double *__restrict__ Eh, double *__restrict__ Fh)
{
    constexpr int NP = 64, CW = 16, DS = CW + 1, PW = 2 * NP, SPAR = NTH / 256;      // SPAR slabs side by side
    static_assert(NTH == 256 || NTH == 1024, "four waves per slab");
    c2 *As = reinterpret_cast<c2 *>(smem_raw);          // [M][Z][64]
    c2 *Ds = As + (size_t)M * Z * 64 + (size_t)(SPAR > 1 ? (threadIdx.x >> 8) : 0) * 64 * DS;      // [64][DS] (this slab's)
    int *Ecol = reinterpret_cast<int *>(As + (size_t)M * Z * 64 + (size_t)SPAR * 64 * DS);  // [Z][64]
    const int tid = threadIdx.x, w = (tid >> 6) & 3, r = tid & 63, cl = 4 * w;
    // every global load of the prologue is issued before the first wait (k_build_LR_ell)
    constexpr int EIT = 1024 / NTH;                     // Z <= 16
    int ecv[EIT];
    #pragma unroll
    for (int it = 0; it < EIT; it++) {
        const int item = tid + it * NTH;
        ecv[it] = (item < Z * 64) ? ell_col[item] : 0;  // [e][row], Np = 64
    }
    uint32_t slot4[4];
    #pragma unroll
    for (int s = 0; s < 4; s++) slot4[s] = *reinterpret_cast<const uint32_t *>(ell_inv + (size_t)r * NP + CW * s + cl);
    assemble_ell<NOPS>(As, ell_val, tab, n, M, M, n_ops, Z, NP, tid, NTH);
    #pragma unroll
    for (int it = 0; it < EIT; it++) {
        const int item = tid + it * NTH;
        if (item < Z * 64) Ecol[item] = ecv[it];
    }
    double cLs[M + 1], cRs[M + 1];
    #pragma unroll
    for (int i = 1; i <= M; i++) { cLs[i] = cw[2 * i + 1]; cRs[i] = cw[2 * i]; }
    __syncthreads();
    // this lane's element of a panel row: column r of L^H / R^H
    const int ocol = (r >> 3) * 16 + (r & 7);
    _Pragma("unroll 1") for (int s = (SPAR > 1 ? (tid >> 8) : 0); s < 4; s += SPAR) {
        const int c0 = CW * s + cl;                     // first of this thread's four columns of L, R
        const uint32_t slots = (s == 0) ? slot4[0] : (s == 1) ? slot4[1] : (s == 2) ? slot4[2] : slot4[3];
        c2 T[M][4], Lacc[4], Racc[4];
        #pragma unroll
        for (int c = 0; c < 4; c++) {
            #pragma unroll
            for (int q = 0; q < M; q++) T[q][c] = (c2){0.0, 0.0};
            const double id = (r == c0 + c) ? 1.0 : 0.0;
            Lacc[c] = (c2){id, 0.0}; Racc[c] = (c2){id, 0.0};
        }
        // source 0 is the identity: A_d I = A_d
        #pragma unroll
        for (int c = 0; c < 4; c++) {
            const int e = (slots >> (8 * c)) & 0xff;
            if (e != 0xff) {
                #pragma unroll
                for (int d = 0; d < M; d++) {
                    const c2 a = As[(d * Z + e) * 64 + r];
                    T[d][c].re += a.re; T[d][c].im += a.im;
                }
            }
        }
        #pragma unroll
        for (int i = 1; i <= M; i++) {
            const double inv = 1.0 / (double)i, cL = cLs[i], cR = cRs[i];
            c2 Di[4];
            #pragma unroll
            for (int c = 0; c < 4; c++) {
                Di[c] = (c2){T[i - 1][c].re * inv, T[i - 1][c].im * inv};
                Lacc[c].re = __builtin_fma(cL, Di[c].re, Lacc[c].re); Lacc[c].im = __builtin_fma(cL, Di[c].im, Lacc[c].im);
                Racc[c].re = __builtin_fma(cR, Di[c].re, Racc[c].re); Racc[c].im = __builtin_fma(cR, Di[c].im, Racc[c].im);
            }
            if (i == M) break;
            // (a wave reads only the columns it wrote: the exchange is among its own lanes, no workgroup barrier)
            wave_lds_fence();
            #pragma unroll
            for (int c = 0; c < 4; c++) Ds[r * DS + cl + c] = Di[c];
            wave_lds_fence();
            _Pragma("unroll 1") for (int e = 0; e < Z; e++) {
                const c2 *src = Ds + Ecol[e * 64 + r] * DS + cl;
                c2 x[4];
                #pragma unroll
                for (int c = 0; c < 4; c++) x[c] = src[c];
                #pragma unroll
                for (int d = 0; d + i < M; d++) {
                    const c2 a = As[(d * Z + e) * 64 + r];
                    #pragma unroll
                    for (int c = 0; c < 4; c++) cfma(T[i + d][c], a, x[c]);
                }
            }
        }
        // rows c0 .. c0+3 of L^H and R^H: element (c, r) = conj(L[r][c]); 8 lanes = 64 contiguous bytes
        #pragma unroll
        for (int c = 0; c < 4; c++) {
            const size_t o = (size_t)(c0 + c) * PW + ocol;
            if (NTH == 1024) {      // (the pre-building launch: nobody reads the panels before the next kernel -- streamed past the L2)
                __builtin_nontemporal_store(Lacc[c].re, Eh + o); __builtin_nontemporal_store(-Lacc[c].im, Eh + o + 8);
                __builtin_nontemporal_store(Racc[c].re, Fh + o); __builtin_nontemporal_store(-Racc[c].im, Fh + o + 8);
            } else {
                Eh[o] = Lacc[c].re; Eh[o + 8] = -Lacc[c].im;
                Fh[o] = Racc[c].re; Fh[o + 8] = -Racc[c].im;
            }
        }
    }
}

// Which workgroups of k_front find their L_n^H, R_n^H built by the launch in front of it (k_tables_front).  k_front is one round
// of up to three workgroups per CU, workgroups b, b + 256, b + 512 on one CU: with nt = 512 + extra time points the CUs
// 0 .. extra-1 hold three, and the launch ends with them.  Pre-built: every third workgroup (b >= 512), the second workgroups
// b = 256 .. 256 + q2 - 1 and the first workgroups b = 0 .. q1 - 1 (q1 <= extra) -- they start with the elimination while the
// rest of their CU builds.  (q2, q1) is the library's choice (qgdk_front_pre_plan: q2 = q1 = extra up to 192 workgroups), measured in
// EXPERIMENTS.md "Round 6": the CUs that hold three then run three eliminations and no build, the others two and two.
struct FrontPre { int extra, q2, q1; };
__host__ __device__ static inline int front_extra(int nt) { const int e = nt - 512; return e < 0 ? 0 : (e > 256 ? 0 : e); }      // (beyond one round: no tail to balance)
__host__ __device__ static inline int front_pre_count(const FrontPre p) { return p.extra + p.q2 + p.q1; }
__host__ __device__ static inline int front_pre_point(int j, const FrontPre p)      // the time point of pre-built workgroup j
{
    return j < p.extra ? 512 + j : (j < p.extra + p.q2 ? 256 + (j - p.extra) : j - p.extra - p.q2);
}
__host__ __device__ static inline bool front_is_prebuilt(int n, const FrontPre p)
{
    return n >= 512 ? p.extra > 0 : (n >= 256 ? n - 256 < p.q2 : n < p.q1);
}
