// Stand-alone timing + correctness of k_inverse_multi<64, NM> (aligned phases, static pivots first) against
// k_inverse_mfma<64>.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I quantumgatedesign.jl_amd/csrc -I include \
//         scripts/ubench/inverse_multi_bench.hip -o scripts/ubench/inverse_multi_bench
//   inverse_multi_bench [nmat=550] [dominant=1]
// dominant=1: L = D + 0.05 noise with |D_ii| ~ 1 (what the dispersive models give); 0: a permuted dominant matrix
// (the static pass must give up and the pivoted pass must produce the inverse).
#include "../../quantumgatedesign.jl_amd/csrc/qgd_k_inverse.hip"
#include <cstdio>
#include <vector>
#include <complex>
#include <random>

static const int NP = 64, PW = 128;
static std::vector<double> Lh;
static std::complex<double> Lat(int n, int r, int c)
{
    const size_t panel = (size_t)NP * PW;
    return {Lh[n * panel + r * PW + (c >> 3) * 16 + (c & 7)], Lh[n * panel + r * PW + (c >> 3) * 16 + 8 + (c & 7)]};
}

int main(int argc, char **argv)
{
    const int nmat = argc > 1 ? atoi(argv[1]) : 550, dominant = argc > 2 ? atoi(argv[2]) : 1;
    const size_t panel = (size_t)NP * PW, pl = (size_t)NP * NP;
    Lh.assign((nmat + 1) * panel, 0.0);
    std::mt19937_64 rng(1);
    std::normal_distribution<double> nd;
    for (int n = 0; n <= nmat; n++)
        for (int r = 0; r < NP; r++)
            for (int c = 0; c < NP; c++) {
                const bool dg = dominant ? (r == c) : (r == (c * 7 + 3) % NP);
                const double re = 0.05 * nd(rng) + (dg ? 0.6 : 0.0), im = 0.05 * nd(rng) + (dg ? 0.8 * ((r & 1) ? 1 : -1) : 0.0);
                Lh[n * panel + r * PW + (c >> 3) * 16 + (c & 7)] = re;
                Lh[n * panel + r * PW + (c >> 3) * 16 + 8 + (c & 7)] = im;
            }
    double *dL, *dT, *dR, *dPr, *dPc; int *dS;
    hipMalloc(&dL, Lh.size() * 8); hipMalloc(&dT, (nmat + 1) * 2 * pl * 8);
    hipMalloc(&dR, Lh.size() * 8); hipMalloc(&dPr, Lh.size() * 8); hipMalloc(&dPc, Lh.size() * 8);
    hipMalloc(&dS, 8); hipMemset(dS, 0, 8);
    hipMemcpy(dL, Lh.data(), Lh.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dR, Lh.data(), Lh.size() * 8, hipMemcpyHostToDevice);   // R := L shifted by one: P_{n-1} = L_n^-1 L_{n-1}
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto check = [&](const char *name) {
        double err = 0, errp = 0, errq = 0;
        std::vector<double> T(2 * pl), Pp(panel), Pq(2 * pl);
        for (int n : {1, nmat / 2, nmat}) {
            hipMemcpy(T.data(), dT + (size_t)n * 2 * pl, 2 * pl * 8, hipMemcpyDeviceToHost);
            hipMemcpy(Pp.data(), dPr + (size_t)(n - 1) * panel, panel * 8, hipMemcpyDeviceToHost);
            hipMemcpy(Pq.data(), dPc + (size_t)(n - 1) * 2 * pl, 2 * pl * 8, hipMemcpyDeviceToHost);
            for (int r = 0; r < NP; r++)
                for (int c = 0; c < NP; c++) {
                    std::complex<double> s = 0, s1 = 0, s2 = 0;
                    for (int k = 0; k < NP; k++) {
                        s += std::complex<double>(T[r * NP + k], T[pl + r * NP + k]) * Lat(n, k, c);
                        s1 += Lat(n, r, k) * std::complex<double>(Pp[k * PW + (c >> 3) * 16 + (c & 7)], Pp[k * PW + (c >> 3) * 16 + 8 + (c & 7)]);
                        s2 += Lat(n, r, k) * std::complex<double>(Pq[k + NP * c], Pq[pl + k + NP * c]);
                    }
                    err = fmax(err, std::abs(s - (r == c ? 1.0 : 0.0)));
                    errp = fmax(errp, std::abs(s1 - Lat(n - 1, r, c))); errq = fmax(errq, std::abs(s2 - Lat(n - 1, r, c)));
                }
        }
        int st[2]; hipMemcpy(st, dS, 8, hipMemcpyDeviceToHost);
        printf("  %-34s max |Linv L - I| = %.2e, |L P - R| = %.2e (panel) %.2e (planes), status %d, re-pivoted workgroups %d\n", name, err, errp, errq, st[0], st[1]);
        hipMemset(dS, 0, 8);
        hipMemset(dT, 0, (nmat + 1) * 2 * pl * 8); hipMemset(dPr, 0, Lh.size() * 8); hipMemset(dPc, 0, Lh.size() * 8);
    };
    auto time_it = [&](const char *name, auto launch) {
        launch(); hipDeviceSynchronize(); check(name);
        for (int i = 0; i < 5; i++) launch();
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < 50; i++) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-36s %4d matrices: %7.2f us per launch (%s)\n", name, nmat, ms / 50 * 1e3, hipGetErrorString(hipGetLastError()));
        hipMemset(dS, 0, 8);
    };
    time_it("k_inverse_mfma<64>", [&]() { hipLaunchKernelGGL((k_inverse_mfma<64>), dim3(nmat), dim3(256), 0, 0, dL, dR, dT, dPr, dPc, 1, dS); });
#define MULTI(NM, ST) time_it("k_inverse_multi<64," #NM "> static=" #ST, [&]() { \
        hipLaunchKernelGGL((k_inverse_multi<64, NM>), dim3((nmat + NM - 1) / NM), dim3(256 * NM), 0, 0, dL, dR, dT, dPr, dPc, 1, nmat + 1, ST, dS, dS + 1); })
#ifdef QGD_INVM_PROFILE
    auto prof = [&](const char *name, auto launch) {
        unsigned long long z[16] = {0}, pr[16];
        hipMemcpyToSymbol(HIP_SYMBOL(g_invm_prof), z, sizeof z);
        launch(); hipDeviceSynchronize();
        hipMemcpyFromSymbol(pr, HIP_SYMBOL(g_invm_prof), sizeof pr);
        unsigned long long tot = 0; for (int i = 0; i < 12; i++) tot += pr[i];
        printf("%s: cycles of workgroup 0 thread 0: load %llu | panels: publish %llu, barrier A %llu, chain %llu, barrier B %llu, update %llu | "
               "to output %llu, stage+barrier %llu, LinvT stores %llu, product %llu, barrier %llu, P out %llu | total %llu\n",
               name, pr[0], pr[1], pr[2], pr[3], pr[4], pr[5], pr[6], pr[7], pr[8], pr[9], pr[10], pr[11], tot);
    };
#define PROF(NM, ST) prof("k_inverse_multi<64," #NM "> static=" #ST, [&]() { \
        hipLaunchKernelGGL((k_inverse_multi<64, NM>), dim3((nmat + NM - 1) / NM), dim3(256 * NM), 0, 0, dL, dR, dT, dPr, dPc, 1, nmat + 1, ST, dS, dS + 1); })
    PROF(1, 1); PROF(3, 1); PROF(3, 0);
    return 0;
#endif
    MULTI(1, 0); MULTI(1, 1); MULTI(2, 0); MULTI(2, 1); MULTI(3, 0); MULTI(3, 1);
    time_it("k_inverse_multi<64,1,3> static=0", [&]() { hipLaunchKernelGGL((k_inverse_multi<64, 1, 3>), dim3(nmat), dim3(256), 0, 0, dL, dR, dT, dPr, dPc, 1, nmat + 1, 0, dS, dS + 1); });
    time_it("k_inverse_multi<64,1,3> static=1", [&]() { hipLaunchKernelGGL((k_inverse_multi<64, 1, 3>), dim3(nmat), dim3(256), 0, 0, dL, dR, dT, dPr, dPc, 1, nmat + 1, 1, dS, dS + 1); });
    return 0;
}
