// qgd_kernels_common.h -- shared device helpers of the gfx950 kernels (see qgd_k_*.hip).
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
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "qgd_device.h"

typedef double d4 __attribute__((ext_vector_type(4)));

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return (int)e_; } while (0)
// hipFuncSetAttribute once per kernel instantiation AND device (it is a host-side driver call and the attribute is
// per device; handles on several GPUs may live in one process).  The record is atomic: two host threads driving
// separate handles at worst both make the call.
#include <atomic>
#define QGD_MAX_DEVICES 16
#define SET_LDS_ONCE(fn, bytes) do { static std::atomic<size_t> done_[QGD_MAX_DEVICES]; int dev_ = 0; (void)hipGetDevice(&dev_); \
    std::atomic<size_t> &d_ = done_[(unsigned)dev_ % QGD_MAX_DEVICES]; \
    if ((size_t)(bytes) > d_.load(std::memory_order_acquire)) { HIPCHK(hipFuncSetAttribute((const void *)(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes))); d_.store((bytes), std::memory_order_release); } } while (0)

// B-operand pair for one 4-deep k-step out of a panel row (LDS or global):
// b1 = [Bre|Bim], b2 = [-Bim|Bre]
__device__ __forceinline__ void panel_b(const double *row16, int c16, double &b1, double &b2)
{
    b1 = row16[c16];
    double t = row16[c16 ^ 8];
    b2 = (c16 < 8) ? -t : t;
}

// sum over each row of 16 lanes, result in lane 15 of the row (DPP inclusive scan, no LDS traffic)
template <int CTRL>
__device__ __forceinline__ double dpp_add_step(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
    return v + __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row16_sum(double v)
{
    v = dpp_add_step<0x111>(v);     // row_shr:1
    v = dpp_add_step<0x112>(v);     // row_shr:2
    v = dpp_add_step<0x114>(v);     // row_shr:4
    v = dpp_add_step<0x118>(v);     // row_shr:8
    return v;
}

// Workgroup barrier that waits for this wave's LDS traffic only.  __syncthreads() is s_waitcnt vmcnt(0) lgkmcnt(0)
// + s_barrier: in the team pipeline below a wave reaches the next barrier right after issuing the global loads of
// its step four steps ahead (and its history stores), so EVERY step barrier waited ~1100 cycles for another team's
// prefetch to land (scripts/ubench/chain_bench.hip -DQGD_CHAIN_PROFILE: 2500 cycles of MFMAs, then 1500 to get
// through the barrier; 370 in the last steps, which prefetch nothing).  All communication between the waves of a
// chain goes through LDS; nothing written to global memory is read again inside the kernel.

__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Exchange through LDS among the lanes of ONE wave (a wave that writes its own columns of a slab and then reads other
// rows of the same columns): the LDS executes a wave's instructions in order, so no s_barrier is needed -- only the
// compiler must keep the order.
__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Buffer-addressed 8-byte load / store: address = descriptor base (SGPRs, wave-uniform) + soff (SGPR or constant)
// + voff (one VGPR byte offset per lane).  No vector ALU instruction is needed to form the address.
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t buffer_of(const double *base)      // base must be wave-uniform
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(base), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ double buffer_load_f64(__amdgpu_buffer_rsrc_t r, int voff, int soff)
{
    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
    return __hiloint2double((int)v.y, (int)v.x);
}
__device__ __forceinline__ void buffer_store_f64(double x, __amdgpu_buffer_rsrc_t r, int voff, int soff)
{
    u32x2 d; d.x = (unsigned)__double2loint(x); d.y = (unsigned)__double2hiint(x);
    __builtin_amdgcn_raw_buffer_store_b64(d, r, voff, soff, 0);
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

// tile shape of the generic GEMM-like kernels (k_level, k_propagator): 4 column groups, 16-deep k chunks
#define LV_NG 4
#define LV_KC 16

#define DISPATCH_NOPS(n_ops, CALL) \
    switch (n_ops) { case 1: CALL(1); break; case 2: CALL(2); break; case 3: CALL(3); break; case 4: CALL(4); break; \
                     default: CALL(-1); break; }
