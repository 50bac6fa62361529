// Microbenchmark: do v_fma_f64 (vector ALU) and v_mfma_f64_16x16x4_f64 (matrix pipe) run CONCURRENTLY on gfx950?
// Both peaks are 78.6 TFLOP/s fp64 on MI355X.  Three kernels with the same loop structure: NM independent MFMA chains only,
// NV independent v_fma_f64 chains only, and both interleaved (per iteration NM MFMAs + NV*R FMAs).  If the pipes overlap, the
// mixed kernel takes max(t_mfma, t_fma), not the sum.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/mfma_valu_coissue.hip -o scripts/ubench/bin/coissue && scripts/ubench/bin/coissue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NM, int NV, int R>
__global__ __launch_bounds__(1024) void k(double *out, int iters, double a0, double b0)
{
    d4 acc[NM > 0 ? NM : 1];
    double v[NV > 0 ? NV : 1];
    for (int i = 0; i < (NM > 0 ? NM : 1); i++) acc[i] = (d4){0, 0, 0, 0};
    for (int i = 0; i < (NV > 0 ? NV : 1); i++) v[i] = threadIdx.x * 1e-3 + i;
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    const double m1 = 0.999999, c1 = 1e-7;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < R; r++) {
#pragma unroll
            for (int i = 0; i < NM; i++) if (r == 0) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < NV; i++) v[i] = __builtin_fma(v[i], m1, c1);
        }
    }
    double s = 0;
    for (int i = 0; i < NM; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < NV; i++) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NM, int NV, int R>
double run(const char *label, int waves_per_cu, int iters)
{
    double *d; hipMalloc(&d, 1 << 24);
    int blocks = 256, threads = 64 * waves_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NM, NV, R><<<blocks, threads>>>(d, 10, 1.0, 1.0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NM, NV, R><<<blocks, threads>>>(d, iters, 1.000001, 0.999999);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double waves = (double)blocks * waves_per_cu;
    const double mf = (double)iters * NM * 2048.0 * waves, vf = (double)iters * NV * R * 128.0 * waves;
    printf("%-34s waves/WG=%d: %.3f ms  MFMA %.1f TFLOP/s  VALU-FMA %.1f TFLOP/s  sum %.1f\n", label, waves_per_cu, ms, mf / ms / 1e9, vf / ms / 1e9,
           (mf + vf) / ms / 1e9);
    hipFree(d);
    return ms;
}

int main()
{
    // per iteration: 4 MFMAs (4 x 64 = 256 pipe cycles) and up to 64 FMAs (64 x 4 = 256 issue cycles)
    for (int w : {4, 8, 16}) {
        run<4, 0, 1>("MFMA only (4 chains)", w, 20000);
        run<0, 8, 8>("FMA only (8 chains x 8)", w, 20000);
        run<4, 8, 8>("mixed 4 MFMA + 64 FMA", w, 20000);
        run<4, 8, 4>("mixed 4 MFMA + 32 FMA", w, 20000);
        run<4, 8, 2>("mixed 4 MFMA + 16 FMA", w, 20000);
    }
    return 0;
}
