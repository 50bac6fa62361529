// A/B of the two Np = 64 batched inverse + propagator kernels in ONE process, interleaved rounds:
//   k_inverse_mfma (4-pivot panels, every wave in every panel, product phase)   vs
//   k_inverse_cb   (column blocks, 16-pivot block steps, [L | R] eliminated together; qgd_inverse_cb.h)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I quantumgatedesign.jl_amd/csrc -I include \
//         scripts/ubench/inverse_cb_bench.hip -o scripts/ubench/bin/inverse_cb_bench
//   inverse_cb_bench [nmat] [data] [pivot_first] [one_round: 0 / 1 forces k_inverse_cb<false / true>]     data 0: I + 0.1 N(0,1) (the conditioning of the cnot3 step matrices)
//                                      data 1: the same with the rows of every 16-row block permuted (in-tile pivoting)
//                                      data 2: unit entries on a permuted diagonal, noise 0.3 (the pivoted attempt finds pivots in the tiles)
//                                      data 3: noise 0.1 (the diagonal attempt is given up for some matrices)
//                                      data 4: I + 1000 (i <-> i + 32) + noise: every pivot inside a diagonal tile meets multipliers of
//                                              1000 -- both column-block attempts are given up, k_inverse_mfma's elimination does it
// Prints median / min us per launch of each kernel and, for a few matrices, max |Linv L - I| and max |L P - R| on both copies of P.
#include "../../quantumgatedesign.jl_amd/csrc/qgd_k_inverse.hip"
extern "C" int qgdk_dense_inverse(const qgdk_ctx *) { return 0; }
extern "C" int qgdk_dense_propagator(const qgdk_ctx *) { return 0; }
#include <cstdio>
#include <vector>
#include <complex>
#include <random>
#include <algorithm>

typedef std::complex<double> cd;

int main(int argc, char **argv)
{
    const int NP = 64, PW = 128, nmat = argc > 1 ? atoi(argv[1]) : 550, data = argc > 2 ? atoi(argv[2]) : 0;
    const size_t panel = (size_t)NP * PW, pl = (size_t)NP * NP;
    std::vector<double> L((nmat + 1) * panel), Rm((nmat + 1) * panel);
    std::mt19937_64 rng(1);
    std::normal_distribution<double> nd;
    auto put = [&](std::vector<double> &A, int n, int r, int c, cd v) {
        A[n * panel + r * PW + (c >> 3) * 16 + (c & 7)] = v.real();
        A[n * panel + r * PW + (c >> 3) * 16 + 8 + (c & 7)] = v.imag();
    };
    auto at = [&](const std::vector<double> &A, int n, int r, int c) {
        return cd(A[n * panel + r * PW + (c >> 3) * 16 + (c & 7)], A[n * panel + r * PW + (c >> 3) * 16 + 8 + (c & 7)]);
    };
    for (int n = 0; n <= nmat; n++)
        for (int r = 0; r < NP; r++)
            for (int c = 0; c < NP; c++) {
                int dr = r;                                  // the row that carries the unit entry of column c
                if (data == 1) dr = (r & ~15) | ((5 * (r & 15) + 3) & 15);
                if (data == 2) dr = (7 * r + 3) % NP;
                const double amp = data == 2 ? 0.3 : data == 3 ? 0.1 : 0.03;      // (data 3: diagonal pivots meet multipliers beyond CB_GROWTH_STATIC in some matrices)
                put(L, n, r, c, cd(amp * nd(rng) + (dr == c ? 1.0 : 0.0) + (data == 4 && c == (r ^ 32) ? 1000.0 : 0.0), amp * nd(rng)));
                put(Rm, n, r, c, cd(nd(rng), nd(rng)));
            }
    double *dL, *dR, *dT[2], *dPr[2], *dPc[2]; int *dS;
    hipMalloc(&dL, L.size() * 8); hipMalloc(&dR, L.size() * 8);
    for (int v = 0; v < 2; v++) { hipMalloc(&dT[v], (nmat + 1) * 2 * pl * 8); hipMalloc(&dPr[v], L.size() * 8); hipMalloc(&dPc[v], L.size() * 8);
        hipMemset(dT[v], 0, (nmat + 1) * 2 * pl * 8); hipMemset(dPr[v], 0, L.size() * 8); hipMemset(dPc[v], 0, L.size() * 8); }
    hipMalloc(&dS, 32); hipMemset(dS, 0, 32);      // [status mfma | status cb | cb: not by the diagonal attempt | cb: by the last resort | cb: pivot first]
    if (argc > 3 && atoi(argv[3])) { const int one = 1; hipMemcpy(dS + 4, &one, 4, hipMemcpyHostToDevice); }
    hipMemcpy(dL, L.data(), L.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dR, Rm.data(), L.size() * 8, hipMemcpyHostToDevice);
    const bool one_round = argc > 4 ? atoi(argv[4]) != 0 : nmat > CB_ONE_ALONE && nmat <= CB_ONE_ROUND;      // (the library's choice, or forced)
    auto launch = [&](int v) {
        if (v == 0) hipLaunchKernelGGL((k_inverse_mfma<64>), dim3(nmat), dim3(256), 0, 0, dL, dR, dT[0], dPr[0], dPc[0], 1, dS);
        else if (one_round) hipLaunchKernelGGL(k_inverse_cb<true>, dim3(nmat), dim3(256), 0, 0, dL, dR, dT[1], dPr[1], dPc[1], 1, dS + 1, dS + 2);
        else hipLaunchKernelGGL(k_inverse_cb<false>, dim3(nmat), dim3(256), 0, 0, dL, dR, dT[1], dPr[1], dPc[1], 1, dS + 1, dS + 2);
    };
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int v = 0; v < 2; v++) { for (int i = 0; i < 3; i++) launch(v); }
    hipDeviceSynchronize();
    printf("launch status: %s\n", hipGetErrorString(hipGetLastError()));
    const int ROUNDS = 12, PER = 10;
    std::vector<float> tms[2];
    for (int rd = 0; rd < ROUNDS; rd++)
        for (int v = 0; v < 2; v++) {
            hipEventRecord(e0);
            for (int i = 0; i < PER; i++) launch(v);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            tms[v].push_back(ms / PER * 1e3f);
        }
    const char *names[2] = {"k_inverse_mfma<64>", "k_inverse_cb      "};
    for (int v = 0; v < 2; v++) {
        std::sort(tms[v].begin(), tms[v].end());
        printf("%s  %d matrices, data %d: median %.2f us, min %.2f us per launch\n", names[v], nmat, data, tms[v][ROUNDS / 2], tms[v][0]);
    }
#ifdef CB_PROFILE
    {
        unsigned long long pr[4][8][8];
        hipMemcpyFromSymbol(pr, HIP_SYMBOL(g_cb_prof), sizeof pr);
        const unsigned long long t0 = pr[0][4][0];
        printf("stamps of workgroup 0 (s_memtime ticks since the kernel's first stamp; about one per shader cycle: an update of 64 MFMAs takes 4500-4700)\n");
        for (int w = 0; w < 4; w++) {
            printf(" wave %d: loaded %6lld start %6lld |", w, (long long)(pr[w][4][0] - t0), (long long)(pr[w][4][1] - t0));
            for (int k = 0; k < 4; k++)
                printf(" k%d: in %6lld panel-end %6lld barrier %6lld updL %6lld updR %6lld |", k, (long long)(pr[w][k][0] - t0), (long long)(pr[w][k][1] - t0),
                       (long long)(pr[w][k][2] - t0), (long long)(pr[w][k][3] - t0), (long long)(pr[w][k][4] - t0));
            printf(" elim-end %6lld out-end %6lld\n", (long long)(pr[w][4][2] - t0), (long long)(pr[w][4][3] - t0));
        }
        for (int sp = 0; sp < 2; sp++)
            printf(" panel %d sub-panel %d: to-LDS+rows %lld | chain %lld | write-back %lld | reads+MFMA %lld | read-back %lld\n", CB_PK, sp, (long long)(pr[0][5 + sp][1] - pr[0][5 + sp][0]),
                   (long long)(pr[0][5 + sp][2] - pr[0][5 + sp][1]), (long long)(pr[0][5 + sp][3] - pr[0][5 + sp][2]), (long long)(pr[0][5 + sp][4] - pr[0][5 + sp][3]), (long long)(pr[0][5 + sp][5] - pr[0][5 + sp][4]));
        // placement of the LAST launch: workgroups per CU, and the waves' SIMDs
        static unsigned long long place[4096][4][4];
        hipMemcpyFromSymbol(place, HIP_SYMBOL(g_cb_place), sizeof place);
        std::vector<int> per_cu(8 * 64, 0);
        unsigned long long tmin = ~0ull;
        for (int b = 0; b < nmat && b < 4096; b++) tmin = std::min(tmin, place[b][0][2]);
        for (int b = 0; b < nmat && b < 4096; b++) {
            const unsigned hw = (unsigned)place[b][0][0], xcc = (unsigned)place[b][0][1] & 15;
            const int cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            per_cu[xcc * 64 + se * 16 + cu]++;      // (SH folded: one SH per SE here)
            if (b < 24 || b % 64 == 0 || b >= nmat - 8) {
                printf(" wg %4d: xcc %u se %d sh %d cu %2d | simd/slot", b, xcc, se, sh, cu);
                for (int w = 0; w < 4; w++) printf(" %u/%u", ((unsigned)place[b][w][0] >> 4) & 3, (unsigned)place[b][w][0] & 15);
                printf(" | start %6lld end %6lld\n", (long long)(place[b][0][2] - tmin), (long long)(place[b][0][3] - tmin));
            }
        }
        int hist[8] = {0}; for (int c : per_cu) hist[std::min(c, 7)]++;
        printf(" CUs with 0..5 workgroups: %d %d %d %d %d %d\n", hist[0], hist[1], hist[2], hist[3], hist[4], hist[5]);
        // completion time by the number of workgroups on the CU
        double sum[8] = {0}; int cnt[8] = {0};
        for (int b = 0; b < nmat && b < 4096; b++) {
            const unsigned hw = (unsigned)place[b][0][0], xcc = (unsigned)place[b][0][1] & 15;
            const int c = per_cu[xcc * 64 + ((hw >> 13) & 7) * 16 + ((hw >> 8) & 15)];
            sum[std::min(c, 7)] += (double)(place[b][0][3] - place[b][0][2]); cnt[std::min(c, 7)]++;
        }
        for (int c = 1; c < 6; c++) if (cnt[c]) printf(" workgroups on a CU with %d: %d, mean in-kernel time %.0f ticks\n", c, cnt[c], sum[c] / cnt[c]);
    }
#endif
    int st[5]; hipMemcpy(st, dS, 20, hipMemcpyDeviceToHost);
    printf("status: mfma %d, cb %d; cb over all launches: %d matrices not done by the diagonal attempt, %d by the last resort; pivot first: %d\n", st[0], st[1], st[2], st[3], st[4]);
    // check a few matrices of each kernel
    const int picks[4] = {1, 2, nmat / 2 + 1, nmat};
    for (int v = 0; v < 2; v++) {
        double e_inv = 0, e_pp = 0, e_pq = 0;
        for (int pi = 0; pi < 4; pi++) {
            const int n = picks[pi];
            std::vector<double> T(2 * pl), Pp(panel), Pq(2 * pl);
            hipMemcpy(T.data(), dT[v] + (size_t)n * 2 * pl, 2 * pl * 8, hipMemcpyDeviceToHost);
            hipMemcpy(Pp.data(), dPr[v] + (size_t)(n - 1) * panel, panel * 8, hipMemcpyDeviceToHost);
            hipMemcpy(Pq.data(), dPc[v] + (size_t)(n - 1) * 2 * pl, 2 * pl * 8, hipMemcpyDeviceToHost);
            for (int r = 0; r < NP; r++)
                for (int c = 0; c < NP; c++) {
                    cd s = 0, s1 = 0, s2 = 0;
                    for (int k = 0; k < NP; k++) {
                        s += cd(T[r * NP + k], T[pl + r * NP + k]) * at(L, n, k, c);
                        s1 += at(L, n, r, k) * cd(Pp[k * PW + (c >> 3) * 16 + (c & 7)], Pp[k * PW + (c >> 3) * 16 + 8 + (c & 7)]);
                        s2 += at(L, n, r, k) * cd(Pq[k + NP * c], Pq[pl + k + NP * c]);
                    }
                    e_inv = fmax(e_inv, std::abs(s - (r == c ? 1.0 : 0.0)));
                    e_pp = fmax(e_pp, std::abs(s1 - at(Rm, n - 1, r, c))); e_pq = fmax(e_pq, std::abs(s2 - at(Rm, n - 1, r, c)));
                }
        }
        printf("%s  max |Linv L - I| = %.2e, max |L P - R| = %.2e (panel) %.2e (planes)\n", names[v], e_inv, e_pp, e_pq);
    }
    return 0;
}
