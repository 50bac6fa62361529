// Microbenchmark: v_mfma_f64_16x16x4_f64 issue rate and dependent latency on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void k(double *out, int iters, double a0, double b0)
{
    d4 acc[NACC];
    for (int i = 0; i < NACC; i++) acc[i] = (d4){0, 0, 0, 0};
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (double)(t1 - t0);
}
template <int NACC>
void run(int waves_per_cu, int iters)
{
    double *d; hipMalloc(&d, 1 << 24);
    int blocks = 256, threads = 64 * waves_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC><<<blocks, threads>>>(d, 10, 1.0, 1.0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NACC><<<blocks, threads>>>(d, iters, 1.000001, 0.999999);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double cyc; hipMemcpy(&cyc, d, 8, hipMemcpyDeviceToHost);
    double nm = (double)iters * NACC;
    double flops = nm * 2048.0 * blocks * waves_per_cu;
    printf("NACC=%d waves/CU=%d: %.1f cycles(memtime ticks)/MFMA/wave, %.2f ms, %.1f TFLOP/s\n", NACC, waves_per_cu,
           cyc / nm, ms, flops / ms / 1e9);
    hipFree(d);
}
int main()
{
    run<1>(4, 20000); run<2>(4, 20000); run<4>(4, 20000); run<8>(4, 10000);
    run<4>(8, 10000); run<4>(16, 10000); run<1>(16, 10000);
    return 0;
}
