// qgd_k_layout.hip -- device panels <-> the reference's column-major host layouts.
//
// The reference hands eval_forward! / discrete_adjoint! the arrays
//   uv_history      [2N, 1+m, 1+nsteps, c]   (src/forward_evolution.jl:33-44)
//   lambda_history  [2N, 1+m, 1+nsteps, c]   (src/eval_grad_discrete_adjoint.jl:107-115)
//   adjoint_forcing [2N, 1+nsteps, c]        (:732-752)
// in Julia (column-major) order; the device keeps states as panels [n][Np][2cp] (qgd_kernels_common.h).
// One workgroup moves one tile of 64 rows x 8 complex columns through LDS: the panel side is read as
// 128-byte row segments, the reference side is written as runs of 64 consecutive doubles (one wave = 512
// contiguous bytes) -- into a device staging buffer, or straight into a host buffer the caller registered
// (qgd_register_host_buffer: the pointer is then a device-visible mapping of pinned host memory).
// HBM-bound: 16 N c bytes read and written per (time point, Taylor index).
#include "qgd_kernels_common.h"

struct LayoutArgs {
    const double *src;      // panels
    double *dst;            // reference layout
    long long src_n, src_j; // panel strides (doubles) per time point and per Taylor index
    long long dst_col, dst_n, dst_j;   // reference-layout strides (doubles)
    int N, Np, c, cp, n0, n_cnt, j_cnt, to_panels;
    int tiles_x, total;     // tiles per (n, j) pair; tiles_x * n_cnt * j_cnt
};

__global__ __launch_bounds__(256) void k_layout(LayoutArgs a)
{
    __shared__ double tile[16][65];
    const int t = threadIdx.x;
    const int ngrp = a.cp >> 3;
    const int PWc = 2 * a.cp;
    // one tile per workgroup, or (a grid smaller than the tile count) a few persistent workgroups that walk the tiles:
    // the form used to write straight into registered host memory beside the adjoint sweep -- the transfer is bound
    // by PCIe, so a handful of CUs saturate it and the rest stay free for the sweep
    for (int tl = blockIdx.x; tl < a.total; tl += gridDim.x) {
        const int bx = tl % a.tiles_x, rest = tl / a.tiles_x;
        const int g = bx % ngrp, rb = bx / ngrp;
        const int n = a.n0 + rest % a.n_cnt, j = rest / a.n_cnt;
        const double *sp = a.src + (size_t)n * a.src_n + (size_t)j * a.src_j;
        double *dp = a.dst + (size_t)n * a.dst_n + (size_t)j * a.dst_j;
        if (!a.to_panels) {
            #pragma unroll
            for (int q = 0; q < 4; q++) {
                const int idx = t + 256 * q, row = idx >> 4, cc = idx & 15;
                const int r = rb * 64 + row;
                tile[cc][row] = (r < a.Np) ? sp[(size_t)r * PWc + 16 * g + cc] : 0.0;
            }
            __syncthreads();
            #pragma unroll
            for (int q = 0; q < 4; q++) {
                const int idx = t + 256 * q, cc = idx >> 6, row = idx & 63;
                const int r = rb * 64 + row, col = 8 * g + (cc & 7);
                if (r < a.N && col < a.c) dp[(size_t)col * a.dst_col + (size_t)(cc >> 3) * a.N + r] = tile[cc][row];
            }
        } else {      // reference layout (device copy) -> panels; padding rows / columns are written as zeros
            #pragma unroll
            for (int q = 0; q < 4; q++) {
                const int idx = t + 256 * q, cc = idx >> 6, row = idx & 63;
                const int r = rb * 64 + row, col = 8 * g + (cc & 7);
                tile[cc][row] = (r < a.N && col < a.c) ? dp[(size_t)col * a.dst_col + (size_t)(cc >> 3) * a.N + r] : 0.0;
            }
            __syncthreads();
            #pragma unroll
            for (int q = 0; q < 4; q++) {
                const int idx = t + 256 * q, row = idx >> 4, cc = idx & 15;
                const int r = rb * 64 + row;
                if (r < a.Np) const_cast<double *>(sp)[(size_t)r * PWc + 16 * g + cc] = tile[cc][row];
            }
        }
        __syncthreads();
    }
}

// panels [n][j][Np][2cp] -> dst[col][n][j][2N] (to_panels = 0) or back (to_panels = 1), for time points
// n0 .. n0+n_cnt-1 and j_cnt Taylor indices
extern "C" int qgdk_layout(const qgdk_ctx *c, const double *panels, long long src_n, long long src_j, double *ref,
                           long long dst_col, long long dst_n, long long dst_j, int n0, int n_cnt, int j_cnt,
                           int to_panels, hipStream_t stream, int max_workgroups)
{
    if (n_cnt <= 0 || j_cnt <= 0) return 0;
    LayoutArgs a;
    a.src = panels; a.dst = ref; a.src_n = src_n; a.src_j = src_j;
    a.dst_col = dst_col; a.dst_n = dst_n; a.dst_j = dst_j;
    a.N = c->N; a.Np = c->Np; a.c = c->c; a.cp = c->cp; a.n0 = n0; a.n_cnt = n_cnt; a.j_cnt = j_cnt; a.to_panels = to_panels;
    a.tiles_x = ((c->Np + 63) / 64) * (c->cp / 8);
    const long long total = (long long)a.tiles_x * n_cnt * j_cnt;
    if (total > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    a.total = (int)total;
    const int grid = (max_workgroups > 0 && max_workgroups < a.total) ? max_workgroups : a.total;
    hipLaunchKernelGGL(k_layout, dim3(grid), dim3(256), 0, stream, a);
    return (int)hipGetLastError();
}

// the singularity flag as a double inside the range a reduction sums (multi-GPU evaluations: every rank then fails together)
__global__ void k_flag_to_scal(const int *__restrict__ status, double *__restrict__ scal)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) scal[3] = *status ? 1.0 : 0.0;
}

extern "C" int qgdk_flag_to_scal(const qgdk_ctx *c)
{
    hipLaunchKernelGGL(k_flag_to_scal, dim3(1), dim3(64), 0, c->stream, c->status, c->scal);
    return (int)hipGetLastError();
}
