// qgd_ell.h -- the sparse-operator (ELL) pieces shared by qgd_k_sparse.hip (step matrices, gradient scalars) and the fused
// front kernel of qgd_k_inverse.hip (qgd_front.h): the 16-byte complex type of the LDS slabs and the assembly of A_d(t_n).
#pragma once
#include "qgd_kernels_common.h"

struct __attribute__((aligned(16))) c2 { double re, im; };      // 16-byte aligned: LDS accesses become ds_read_b128 / ds_write_b128

__device__ __forceinline__ void cfma(c2 &acc, const c2 a, const c2 x)
{
    acc.re = __builtin_fma(a.re, x.re, acc.re); acc.re = __builtin_fma(-a.im, x.im, acc.re);
    acc.im = __builtin_fma(a.re, x.im, acc.im); acc.im = __builtin_fma(a.im, x.re, acc.im);
}

// assemble A_d(t_n) = K - iS, d = 0..nd-1, over the union pattern into LDS:
// As[(d*Z + e)*64 + r] = (K, -S).  One (entry, row) pair per thread: the operator values are
// fetched once and combined with the coefficients of every derivative order.
template <int NOPS = -1>       // NOPS >= 0: the operator count at compile time (no exec-masked branch around each load)
__device__ __forceinline__ void assemble_ell(c2 *As, const double *__restrict__ ell_val,
                                             const double *__restrict__ tab, int n, int m, int nd,
                                             int n_ops, int Z, int Np, int tid, int nth)
{
    const size_t per = (size_t)Z * Np;
    for (int pair = tid; pair < Z * 64; pair += nth) {
        const int r = pair & 63, e = pair >> 6;
        const bool live = r < Np;
        const size_t at = (size_t)e * Np + (live ? r : 0);
        double kv[QGD_MAX_OPS_DEV + 1], sv[QGD_MAX_OPS_DEV + 1];
        kv[0] = live ? ell_val[at] : 0.0;
        sv[0] = live ? ell_val[per + at] : 0.0;
        #pragma unroll
        for (int o = 0; o < NOPS_LIM(NOPS); o++) {
            const bool on = live && NOPS_ON(NOPS, o, n_ops);
            kv[o + 1] = on ? ell_val[(size_t)(2 + 2 * o) * per + at] : 0.0;
            sv[o + 1] = on ? ell_val[(size_t)(3 + 2 * o) * per + at] : 0.0;
        }
        for (int d = 0; d < nd; d++) {
            const double *t = tab + (((size_t)n * (m + 1) + d) * n_ops) * 2;
            double K = (d == 0) ? kv[0] : 0.0, S = (d == 0) ? sv[0] : 0.0;
            #pragma unroll
            for (int o = 0; o < NOPS_LIM(NOPS); o++) {
                if (NOPS_ON(NOPS, o, n_ops)) { K = __builtin_fma(t[2 * o + 1], kv[o + 1], K); S = __builtin_fma(t[2 * o], sv[o + 1], S); }
            }
            As[(d * Z + e) * 64 + r] = (c2){K, -S};
        }
    }
}

