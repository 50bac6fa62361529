// Stand-alone timing + phase profile of the ROUND-4 Np=64 batched inverse kernel (4-pivot panels; frozen in
// inverse_r4_kernel.h with its knock-out switches -- the library's kernels are in csrc/qgd_k_inverse.hip and
// csrc/qgd_inverse_cb.h, their A/B is inverse_cb_bench.hip).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I quantumgatedesign.jl_amd/csrc -I include \
//         scripts/ubench/inverse_bench.hip -o scripts/ubench/bin/inverse_bench [-DQGD_INV_PROFILE] [-DQGD_INV_KO_CHAIN ...]
// Prints the mean launch time over 50 launches of nmat matrices, the max |Linv*L - I|, and (with
// -DQGD_INV_PROFILE) the cycles one workgroup spent in each phase of the blocked elimination.
#include "inverse_r4_kernel.h"
#define KERNEL k_inverse_r4
#define STR2(x) #x
#define STR(x) STR2(x)
#include <cstdio>
#include <vector>
#include <complex>
#include <random>

int main(int argc, char **argv)
{
    const int NP = 64, PW = 128, nmat = argc > 1 ? atoi(argv[1]) : 549;
    const size_t panel = (size_t)NP * PW, pl = (size_t)NP * NP;
    std::vector<double> L((nmat + 1) * panel);
    std::mt19937_64 rng(1);
    std::normal_distribution<double> nd;
    for (int n = 0; n <= nmat; n++)
        for (int r = 0; r < NP; r++)
            for (int c = 0; c < NP; c++) {
                const double re = 0.3 * nd(rng) + (r == (c * 7 + 3) % NP ? 1.0 : 0.0), im = 0.3 * nd(rng);
                L[n * panel + r * PW + (c >> 3) * 16 + (c & 7)] = re;
                L[n * panel + r * PW + (c >> 3) * 16 + 8 + (c & 7)] = im;
            }
    double *dL, *dA, *dT, *dR, *dPr, *dPc; int *dS; unsigned long long *dprof;
    hipMalloc(&dL, L.size() * 8); hipMalloc(&dA, (nmat + 1) * 2 * pl * 8); hipMalloc(&dT, (nmat + 1) * 2 * pl * 8);
    hipMalloc(&dR, L.size() * 8); hipMalloc(&dPr, L.size() * 8); hipMalloc(&dPc, L.size() * 8);
    hipMalloc(&dS, 4); hipMemset(dS, 0, 4); hipMalloc(&dprof, 64 * 8); hipMemset(dprof, 0, 64 * 8);
    hipMemcpy(dL, L.data(), L.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dR, L.data(), L.size() * 8, hipMemcpyHostToDevice);   // R := L shifted by one: P_{n-1} = L_n^-1 L_{n-1}
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto launch = [&]() {
#ifdef QGD_INV_PROFILE
        hipLaunchKernelGGL((k_inverse_r4<64>), dim3(nmat), dim3(256), 0, 0, dL, dR, dT, dPr, dPc, 1, dS, dprof);
#else
        hipLaunchKernelGGL((KERNEL<64>), dim3(nmat), dim3(256), 0, 0, dL, dR, dT, dPr, dPc, 1, dS);
#endif
    };
    for (int i = 0; i < 5; i++) launch();
    hipDeviceSynchronize();
    hipMemset(dprof, 0, 64 * 8);
    hipEventRecord(e0);
    for (int i = 0; i < 50; i++) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf(STR(KERNEL) "<64>: %d matrices, %.2f us per launch (%s)\n", nmat, ms / 50 * 1e3, hipGetErrorString(hipGetLastError()));
    // check matrix 1 (blockIdx 0)
    std::vector<double> T(2 * pl);
    hipMemcpy(T.data(), dT + 2 * pl, 2 * pl * 8, hipMemcpyDeviceToHost);
    double err = 0;
    for (int r = 0; r < NP; r++)
        for (int c = 0; c < NP; c++) {
            std::complex<double> s = 0;
            for (int k = 0; k < NP; k++) {
                std::complex<double> a(T[r * NP + k], T[pl + r * NP + k]);
                std::complex<double> b(L[panel + k * PW + (c >> 3) * 16 + (c & 7)], L[panel + k * PW + (c >> 3) * 16 + 8 + (c & 7)]);
                s += a * b;
            }
            err = fmax(err, std::abs(s - (r == c ? 1.0 : 0.0)));
        }
    // P_0 = L_1^-1 R_0 with R_0 = L_0: check L_1 P_0 = L_0 on the panel copy and the plane copy
    std::vector<double> Pp(panel), Pq(2 * pl);
    hipMemcpy(Pp.data(), dPr, panel * 8, hipMemcpyDeviceToHost); hipMemcpy(Pq.data(), dPc, 2 * pl * 8, hipMemcpyDeviceToHost);
    double errp = 0, errq = 0;
    auto Lat = [&](int n, int r, int c) { return std::complex<double>(L[n * panel + r * PW + (c >> 3) * 16 + (c & 7)], L[n * panel + r * PW + (c >> 3) * 16 + 8 + (c & 7)]); };
    for (int r = 0; r < NP; r++)
        for (int c = 0; c < NP; c++) {
            std::complex<double> s1 = 0, s2 = 0;
            for (int k = 0; k < NP; k++) {
                s1 += Lat(1, r, k) * std::complex<double>(Pp[k * PW + (c >> 3) * 16 + (c & 7)], Pp[k * PW + (c >> 3) * 16 + 8 + (c & 7)]);
                s2 += Lat(1, r, k) * std::complex<double>(Pq[k + NP * c], Pq[pl + k + NP * c]);
            }
            errp = fmax(errp, std::abs(s1 - Lat(0, r, c))); errq = fmax(errq, std::abs(s2 - Lat(0, r, c)));
        }
    printf("max |L1 P0 - R0| = %.2e (panel), %.2e (planes)\n", errp, errq);
    int st; hipMemcpy(&st, dS, 4, hipMemcpyDeviceToHost);
    printf("max |Linv L - I| = %.2e, status %d\n", err, st);
#ifdef QGD_INV_PROFILE
    unsigned long long prof[64]; hipMemcpy(prof, dprof, sizeof prof, hipMemcpyDeviceToHost);
    const char *names[] = {"load", "1 publish", "2 panel GJ (one wave)", "3 pivot rows", "4 mfma+cols", "output + propagator"};
    for (int i = 0; i < 6; i++) printf("  %-22s %10.0f cycles (clock64 ticks) per matrix\n", names[i], (double)prof[i] / 50.0);
#endif
    return 0;
}
