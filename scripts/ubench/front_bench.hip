// Gate (ii) of the round-6 plan: the fused front kernel (k_front: build of L_n^H, R_n^H + column-block elimination of
// [L_n^H | R_n^H] in one workgroup per time point, csrc/qgd_front.h) beside the two launches it replaces
// (k_build_LR_ell<4,8,3> + k_inverse_cb), one process, interleaved rounds.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I quantumgatedesign.jl_amd/csrc -I include \
//         scripts/ubench/front_bench.hip -o scripts/ubench/bin/front_bench
//   front_bench [nt] [one: 0 / 1 forces the instantiation of both column-block kernels] [pre: 1 = the step matrices of front_is_prebuilt() by a launch of their own in front]
//               [pivot_first: 1 = the fused kernel skips the diagonal attempt] [kick: drift entries (i, i +- 16) of this size -> last resort]
// Operators: the banded pattern of a (4,4,4) qudit register -- offsets 0, +-1, +-4, +-16 where the sub-index allows it, slots
// ordered by diagonal (qgd_host_alloc.cpp) -- with random values: drift on the diagonal, three control operators.
// Prints median / min us per launch (pair) and, for a few time points, max |L^H X - I| and |L^H Y - R^H| of the fused kernel's
// outputs against the step matrices the separate build kernel wrote.
#include "../../quantumgatedesign.jl_amd/csrc/qgd_k_inverse.hip"
#include "../../quantumgatedesign.jl_amd/csrc/qgd_k_sparse.hip"
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
    constexpr int NP = 64, PW = 128, M = 4, NOPS = 3, Z = 7;
    const int nt = argc > 1 ? atoi(argv[1]) : 551;
    const size_t panel = (size_t)NP * PW, pl = (size_t)NP * NP;
    std::mt19937_64 rng(3);
    std::uniform_real_distribution<double> ud(-1.0, 1.0);
    // the pattern
    const int offs[Z] = {0, 1, -1, 4, -4, 16, -16};
    std::vector<int32_t> ell_col(Z * NP);
    std::vector<uint8_t> ell_inv(NP * NP, 0xff);
    std::vector<char> present(Z * NP, 0);
    for (int r = 0; r < NP; r++) {
        const int a = r & 3, b = (r >> 2) & 3, c = r >> 4;
        const bool ok[Z] = {true, a < 3, a > 0, b < 3, b > 0, c < 3, c > 0};
        for (int e = 0; e < Z; e++) {
            ell_col[e * NP + r] = (r + offs[e] + NP) % NP;
            present[e * NP + r] = ok[e];
            if (ok[e]) ell_inv[r * NP + r + offs[e]] = (uint8_t)e;
        }
    }
    // values: planes K_sys, S_sys, (Asym_o, Sym_o) x 3; K antisymmetric, S symmetric
    std::vector<double> ell_val((size_t)(2 + 2 * NOPS) * Z * NP, 0.0);
    auto val = [&](int plane, int e, int r) -> double & { return ell_val[((size_t)plane * Z + e) * NP + r]; };
    for (int r = 0; r < NP; r++) val(1, 0, r) = 0.8 * ud(rng);                 // drift: diagonal of S
    const double kick = argc > 5 ? atof(argv[5]) : 0.0;      // > 0: large entries (i, i +- 16) in the drift, outside the diagonal tiles: both column-block attempts are given up
    if (kick > 0) for (int r = 0; r + 16 < NP; r++) { val(0, 5, r) = kick; val(0, 6, r + 16) = -kick; }
    for (int o = 0; o < NOPS; o++)
        for (int e = 1; e < Z; e += 2)                                          // pairs (+off, -off)
            for (int r = 0; r < NP; r++)
                if (present[e * NP + r]) {
                    const int r2 = r + offs[e];
                    const double k = 0.5 * ud(rng), s = 0.5 * ud(rng);
                    val(2 + 2 * o, e, r) = k; val(2 + 2 * o, e + 1, r2) = -k;
                    val(3 + 2 * o, e, r) = s; val(3 + 2 * o, e + 1, r2) = s;
                }
    std::vector<double> tab((size_t)nt * (M + 1) * NOPS * 2);
    for (auto &t : tab) t = 0.3 * ud(rng);
    const double dt = 0.5;
    auto fact = [](int n) { double f = 1; for (int i = 2; i <= n; i++) f *= i; return f; };
    std::vector<double> cw(2 * (M + 1));
    for (int j = 0; j <= M; j++) {
        const double cj = fact(M) * fact(2 * M - j) / (fact(2 * M) * fact(M - j));
        cw[2 * j] = cj * pow(dt, j); cw[2 * j + 1] = cj * pow(-dt, j);
    }
    int32_t *d_col; uint8_t *d_inv; double *d_val, *d_tab, *d_cw;
    hipMalloc(&d_col, ell_col.size() * 4); hipMalloc(&d_inv, ell_inv.size()); hipMalloc(&d_val, ell_val.size() * 8);
    hipMalloc(&d_tab, tab.size() * 8); hipMalloc(&d_cw, cw.size() * 8);
    hipMemcpy(d_col, ell_col.data(), ell_col.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_inv, ell_inv.data(), ell_inv.size(), hipMemcpyHostToDevice);
    hipMemcpy(d_val, ell_val.data(), ell_val.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(d_tab, tab.data(), tab.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(d_cw, cw.data(), cw.size() * 8, hipMemcpyHostToDevice);
    // separate kernels: L, R, then LinvT / Pr / Pc;  fused: Eh, Fh, then its own LinvT / Pr / Pc
    double *dL, *dR, *dE, *dF, *dT[2], *dPr[2], *dPc[2]; int *dS;
    const size_t msz = (size_t)(nt + 1) * panel * 8;
    hipMalloc(&dL, msz); hipMalloc(&dR, msz); hipMalloc(&dE, msz); hipMalloc(&dF, msz);
    for (int v = 0; v < 2; v++) { hipMalloc(&dT[v], msz); hipMalloc(&dPr[v], msz); hipMalloc(&dPc[v], msz);
        hipMemset(dT[v], 0, msz); hipMemset(dPr[v], 0, msz); hipMemset(dPc[v], 0, msz); }
    hipMalloc(&dS, 64); hipMemset(dS, 0, 64);
    if (argc > 4 && atoi(argv[4])) { const int onei = 1; hipMemcpy(dS + 11, &onei, 4, hipMemcpyHostToDevice); }      // fused kernel: skip the diagonal attempt (fallbacks[2])

    const int n_pre = argc > 3 ? atoi(argv[3]) : 0;       // 1: the time points front_is_prebuilt() are built by a launch in front of k_front (the library: by k_tables_front)
    FrontPre pre{0, 0, 0};
    if (n_pre) { pre.extra = front_extra(nt); pre.q2 = pre.extra; }
    const bool one = argc > 2 ? atoi(argv[2]) != 0 : (nt > CB_ONE_ALONE && nt <= CB_ONE_ROUND);
    const size_t shm_b = lds_build_ell(M, Z, 8), shm_f = front_lds(M, Z);
    hipFuncSetAttribute((const void *)k_build_LR_ell<M, 8, NOPS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm_b);
    hipFuncSetAttribute((const void *)k_front<M, NOPS, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm_f);
    hipFuncSetAttribute((const void *)k_front_pre<M, NOPS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm_f);
    hipFuncSetAttribute((const void *)k_front<M, NOPS, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm_f);
    printf("LDS: build %zu, front %zu bytes; instantiation %s\n", shm_b, shm_f, one ? "<true>" : "<false>");
    auto launch = [&](int v) {
        if (v == 0) {
            hipLaunchKernelGGL((k_build_LR_ell<M, 8, NOPS>), dim3(nt, 2), dim3(512), shm_b, 0, d_col, d_inv, d_val, d_tab, dL, dR, d_cw, NP, NOPS, Z);
            if (one) hipLaunchKernelGGL(k_inverse_cb<true>, dim3(nt - 1), dim3(256), 0, 0, dL, dR, dT[0], dPr[0], dPc[0], 1, dS, dS + 1);
            else hipLaunchKernelGGL(k_inverse_cb<false>, dim3(nt - 1), dim3(256), 0, 0, dL, dR, dT[0], dPr[0], dPc[0], 1, dS, dS + 1);
        } else {
            if (n_pre) hipLaunchKernelGGL((k_front_pre<M, NOPS>), dim3(front_pre_count(pre)), dim3(256), shm_f, 0, d_col, d_inv, d_val, d_tab, d_cw, NOPS, Z, dE, dF, pre);
            if (one) hipLaunchKernelGGL((k_front<M, NOPS, true>), dim3(nt), dim3(256), shm_f, 0, d_col, d_inv, d_val, d_tab, d_cw, NOPS, Z, dE, dF, dT[1], dPr[1], dPc[1], dS + 8, dS + 9, pre);
            else hipLaunchKernelGGL((k_front<M, NOPS, false>), dim3(nt), dim3(256), shm_f, 0, d_col, d_inv, d_val, d_tab, d_cw, NOPS, Z, dE, dF, dT[1], dPr[1], dPc[1], dS + 8, dS + 9, pre);
        }
    };
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int v = 0; v < 2; v++) for (int i = 0; i < 3; i++) launch(v);
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
    const char *names[2] = {"k_build_LR_ell + k_inverse_cb", "k_front                      "};
    for (int v = 0; v < 2; v++) {
        std::sort(tms[v].begin(), tms[v].end());
        printf("%s  %d time points: median %.2f us, min %.2f us per evaluation front\n", names[v], nt, tms[v][ROUNDS / 2], tms[v][0]);
    }
#ifdef CB_PROFILE
    {   // the LAST launch was the fused kernel: per workgroup build start / end, elimination start / end (device wall clock, 10 ns), and its CU
        static unsigned long long place[4096][4][4], fp[4096][4];
        hipMemcpyFromSymbol(place, HIP_SYMBOL(g_cb_place), sizeof place);
        hipMemcpyFromSymbol(fp, HIP_SYMBOL(g_front_prof), sizeof fp);
        std::vector<int> per_cu(8 * 64, 0), cu_of(nt);
        unsigned long long tmin = ~0ull, tmax = 0;
        for (int b = 0; b < nt && b < 4096; b++) { tmin = std::min(tmin, fp[b][0]); tmax = std::max(tmax, fp[b][3]); }
        for (int b = 0; b < nt && b < 4096; b++) {
            const unsigned hw = (unsigned)place[b][0][0], xcc = (unsigned)place[b][0][1] & 15;
            cu_of[b] = xcc * 64 + ((hw >> 13) & 7) * 16 + ((hw >> 8) & 15);
            per_cu[cu_of[b]]++;
        }
        printf(" first start to last end: %.2f us\n", (tmax - tmin) * 0.01);
        double sb[8] = {0}, se[8] = {0}, en[8] = {0}, st0[8] = {0}, enmax[8] = {0}; int cnt[8] = {0};
        for (int b = 0; b < nt && b < 4096; b++) {
            const int c = std::min(per_cu[cu_of[b]], 7);
            st0[c] += (fp[b][0] - tmin) * 0.01; sb[c] += (fp[b][1] - fp[b][0]) * 0.01; se[c] += (fp[b][3] - fp[b][2]) * 0.01;
            en[c] += (fp[b][3] - tmin) * 0.01; enmax[c] = fmax(enmax[c], (fp[b][3] - tmin) * 0.01); cnt[c]++;
            if (b < 4 || b % 128 == 0 || b >= nt - 3)
                printf(" wg %4d (cu %3d, %d on it): start %6.2f build-end %6.2f elim-start %6.2f end %6.2f us\n", b, cu_of[b], per_cu[cu_of[b]], (fp[b][0] - tmin) * 0.01,
                       (fp[b][1] - tmin) * 0.01, (fp[b][2] - tmin) * 0.01, (fp[b][3] - tmin) * 0.01);
        }
        for (int c = 1; c < 6; c++) if (cnt[c]) printf(" workgroups on a CU with %d: %d; mean start %.2f, build %.2f, elimination %.2f, end %.2f (latest %.2f) us\n", c, cnt[c], st0[c] / cnt[c], sb[c] / cnt[c], se[c] / cnt[c], en[c] / cnt[c], enmax[c]);
    }
#endif
    int st[16]; hipMemcpy(st, dS, 64, hipMemcpyDeviceToHost);
    printf("status: separate %d (not diagonal %d, last resort %d); fused %d (not diagonal %d, last resort %d)\n", st[0], st[1], st[2], st[8], st[9], st[10]);
    // check: L, R of the separate build against the fused kernel's X = L^-H (row-major planes), Y = S^H (panel, planes)
    auto at = [&](const std::vector<double> &A, int r, int c) {
        return cd(A[r * PW + (c >> 3) * 16 + (c & 7)], A[r * PW + (c >> 3) * 16 + 8 + (c & 7)]);
    };
    const int picks[4] = {0, 1, nt / 2, nt - 1};
    double e_x = 0, e_yp = 0, e_yc = 0, e_e = 0, cond = 0;
    for (int pi = 0; pi < 4; pi++) {
        const int n = picks[pi];
        std::vector<double> Lh(panel), Rh(panel), E(panel), X(2 * pl), Sp(panel), Sc(2 * pl);
        hipMemcpy(Lh.data(), dL + (size_t)n * panel, panel * 8, hipMemcpyDeviceToHost);
        hipMemcpy(Rh.data(), dR + (size_t)n * panel, panel * 8, hipMemcpyDeviceToHost);
        hipMemcpy(E.data(), dE + (size_t)n * panel, panel * 8, hipMemcpyDeviceToHost);
        hipMemcpy(X.data(), dT[1] + (size_t)n * 2 * pl, 2 * pl * 8, hipMemcpyDeviceToHost);
        hipMemcpy(Sp.data(), dPr[1] + (size_t)n * panel, panel * 8, hipMemcpyDeviceToHost);
        hipMemcpy(Sc.data(), dPc[1] + (size_t)n * 2 * pl, 2 * pl * 8, hipMemcpyDeviceToHost);
        for (int r = 0; r < NP; r++)
            for (int c = 0; c < NP; c++) {
                if (!n_pre) e_e = fmax(e_e, std::abs(at(E, r, c) - std::conj(at(Lh, c, r))));      // (a relayout overwrites E)
                cd s = 0, s1 = 0, s2 = 0;
                for (int k = 0; k < NP; k++) {
                    s += std::conj(at(Lh, k, r)) * cd(X[k * NP + c], X[pl + k * NP + c]);            // L^H X
                    s1 += at(Sp, r, k) * at(Lh, k, c);                                               // S L, S from the row-major panel
                    s2 += cd(Sc[r + NP * k], Sc[pl + r + NP * k]) * at(Lh, k, c);                    // S L, S from the column-major planes
                }
                e_x = fmax(e_x, std::abs(s - (r == c ? 1.0 : 0.0)));
                e_yp = fmax(e_yp, std::abs(s1 - at(Rh, r, c))); e_yc = fmax(e_yc, std::abs(s2 - at(Rh, r, c)));
                cond = fmax(cond, std::abs(cd(X[r * NP + c], X[pl + r * NP + c])));
            }
    }
    printf("fused: max |E - L^H| = %.2e, |L^H X - I| = %.2e, |S L - R| = %.2e (panel) %.2e (planes); max |X| entry %.2f\n", e_e, e_x, e_yp, e_yc, cond);
    return 0;
}
