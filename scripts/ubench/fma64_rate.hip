// Issue rate of v_fma_f64 on gfx950 by operand pattern (one wave; s_memtime ticks per FMA):
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/fma64_rate.hip -o scripts/ubench/bin/fma64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void k(double *out, unsigned long long *t, double a0, double b0)
{
    double acc[16], a[4], b[4];
    for (int i = 0; i < 16; i++) acc[i] = threadIdx.x + i;
    for (int i = 0; i < 4; i++) { a[i] = a0 + i + threadIdx.x * 1e-3; b[i] = b0 - i + threadIdx.x * 1e-3; }
    const double sa = __builtin_amdgcn_readfirstlane((int)a0) * 0.5, sb = b0;     // wave-uniform (SGPR) operands
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    #pragma unroll 1
    for (int it = 0; it < 256; it++) {
        #pragma unroll
        for (int r = 0; r < 4; r++)
            #pragma unroll
            for (int i = 0; i < 16; i++) {
                if (MODE == 0) acc[i] = __builtin_fma(a[i & 3], b[(i >> 2) & 3], acc[i]);          // 3 VGPR operands, 16 chains
                if (MODE == 1) acc[i] = __builtin_fma(sa, b[i & 3], acc[i]);                      // one SGPR operand
                if (MODE == 2) acc[i & 3] = __builtin_fma(a[i & 3], b[(i >> 2) & 3], acc[i & 3]);    // 4 chains
                if (MODE == 3) acc[i & 1] = __builtin_fma(a[i & 3], b[(i >> 2) & 3], acc[i & 1]);    // 2 chains
                if (MODE == 4) acc[i] = acc[i] * a[i & 3];                                        // v_mul_f64
                if (MODE == 5) acc[i] = acc[i] + a[i & 3];                                        // v_add_f64
            }
        asm volatile("" ::: "memory");
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0; for (int i = 0; i < 16; i++) s += acc[i];
    out[threadIdx.x] = s + sb;
    if (threadIdx.x == 0) t[MODE] = t1 - t0;
}
int main()
{
    double *out; unsigned long long *t, h[8];
    hipMalloc(&out, 64 * 8); hipMalloc(&t, 64);
    for (int rep = 0; rep < 2; rep++) {
        k<0><<<1, 64>>>(out, t, 1.0, 2.0); k<1><<<1, 64>>>(out, t, 1.0, 2.0); k<2><<<1, 64>>>(out, t, 1.0, 2.0);
        k<3><<<1, 64>>>(out, t, 1.0, 2.0); k<4><<<1, 64>>>(out, t, 1.0, 2.0); k<5><<<1, 64>>>(out, t, 1.0, 2.0);
        hipDeviceSynchronize();
    }
    hipMemcpy(h, t, 64, hipMemcpyDeviceToHost);
    const char *nm[6] = {"fma 3 VGPR, 16 chains", "fma SGPR x VGPR + acc", "fma 4 chains", "fma 2 chains", "mul 16 chains", "add 16 chains"};
    for (int m = 0; m < 6; m++) printf("%-24s %.2f ticks per instruction\n", nm[m], (double)h[m] / (256.0 * 64));
    return 0;
}
