// Stand-alone check + timing of the chain kernels (k_chain_fast) for one launch shape.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I quantumgatedesign.jl_amd/csrc -I include \
//         scripts/ubench/chain_bench.hip -o scripts/ubench/chain_bench
//   chain_bench MODE nblocks blen [Np=64]
// Random step matrices (scaled to keep the products O(1)), random forcing; compares every output of
// the launch with a host evaluation and prints the time per launch.
#include "../../quantumgatedesign.jl_amd/csrc/qgd_k_chain.hip"
extern "C" int qgdk_dense_lambda(const qgdk_ctx *) { return 0; }   // lives in qgd_k_dense.hip; not exercised here
#include <cstdio>
#include <vector>
#include <complex>
#include <random>
typedef std::complex<double> cd;

int main(int argc, char **argv)
{
    const int MODE = argc > 1 ? atoi(argv[1]) : 1, nblocks = argc > 2 ? atoi(argv[2]) : 64, blen = argc > 3 ? atoi(argv[3]) : 9;
    const int inplace = argc > 5 ? atoi(argv[5]) : 0;   // 1: the start panel lives in the output array (slot S), as in the level-2 chains
    const int Np = argc > 4 ? atoi(argv[4]) : 64, cp = 8, S = nblocks * blen - (blen > 2 ? 2 : 0);   // last block shorter
    const size_t pl = (size_t)Np * Np, hstep = (size_t)Np * 2 * cp;
    std::mt19937_64 rng(7); std::normal_distribution<double> nd;
    std::vector<cd> P((size_t)S * pl);
    for (auto &z : P) z = cd(nd(rng), nd(rng)) * (0.7 / std::sqrt((double)Np));
    // device layouts: planes (col-major re, im) and panels (row-major, groups of 8 re + 8 im)
    std::vector<double> Pc((size_t)S * 2 * pl), Pr((size_t)S * 2 * pl);
    for (int n = 0; n < S; n++) for (int r = 0; r < Np; r++) for (int c = 0; c < Np; c++) {
        const cd z = P[n * pl + r * Np + c];
        Pc[(size_t)n * 2 * pl + r + (size_t)Np * c] = z.real(); Pc[(size_t)n * 2 * pl + pl + r + (size_t)Np * c] = z.imag();
        Pr[(size_t)n * 2 * pl + (size_t)r * 2 * Np + (c >> 3) * 16 + (c & 7)] = z.real();
        Pr[(size_t)n * 2 * pl + (size_t)r * 2 * Np + (c >> 3) * 16 + 8 + (c & 7)] = z.imag();
    }
    auto panel_at = [&](int row, int col) { return (size_t)row * 2 * cp + (col >> 3) * 16 + (col & 7); };
    std::vector<double> start((size_t)nblocks * hstep), forcing((size_t)(S + 1) * hstep), out((size_t)(S + 1) * hstep, 0.0);
    for (auto &v : start) v = nd(rng);
    for (auto &v : forcing) v = nd(rng);
    double *dPc, *dPr, *dstart, *dforcing, *dout, *dPiC, *dPiR, *dphi;
    hipMalloc(&dPc, Pc.size() * 8); hipMalloc(&dPr, Pr.size() * 8); hipMalloc(&dstart, start.size() * 8);
    hipMalloc(&dforcing, forcing.size() * 8); hipMalloc(&dout, out.size() * 8);
    hipMalloc(&dPiC, (size_t)nblocks * 2 * pl * 8); hipMalloc(&dPiR, (size_t)nblocks * 2 * pl * 8); hipMalloc(&dphi, (size_t)nblocks * hstep * 8);
    hipMemcpy(dPc, Pc.data(), Pc.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dPr, Pr.data(), Pr.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dstart, start.data(), start.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dforcing, forcing.data(), forcing.size() * 8, hipMemcpyHostToDevice);
    hipMemset(dout, 0, out.size() * 8);
    if (inplace) hipMemcpy(dout + (size_t)S * hstep, start.data(), hstep * 8, hipMemcpyHostToDevice);
    ChainArgs a{};
    a.Np = Np; a.cp = cp; a.S = S; a.Pmat = (MODE >= 2) ? dPr : dPc; a.start = inplace ? dout + (size_t)S * hstep : dstart; a.start_stride = inplace ? 0 : (long long)hstep;
    a.out = dout; a.forcing = dforcing; a.PiC = dPiC; a.PiR = dPiR; a.phi = dphi; a.nblocks = nblocks; a.blen = blen;
    a.ngroups = (MODE == 0) ? Np / 8 : cp / 8;
    auto launch = [&]() { return MODE == 0 ? launch_chain<0>(a, 0) : MODE == 1 ? launch_chain<1>(a, 0) : MODE == 2 ? launch_chain<2>(a, 0) : launch_chain<3>(a, 0); };
    for (int i = 0; i < 3; i++) launch();
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int i = 0; i < 20; i++) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("mode %d, %d blocks x %d steps, Np %d: %.2f us per launch = %.3f us per step (%s)\n", MODE, nblocks, blen, Np,
           ms / 20 * 1e3, ms / 20 * 1e3 / blen, hipGetErrorString(hipGetLastError()));
    // host reference
    const int C = (MODE == 0) ? Np : 8;
    double err = 0.0, ref_max = 0.0;
    std::vector<double> hout(out.size()), hphi((size_t)nblocks * hstep), hPiR((size_t)nblocks * 2 * pl);
    hipMemcpy(hout.data(), dout, out.size() * 8, hipMemcpyDeviceToHost);
    hipMemcpy(hphi.data(), dphi, hphi.size() * 8, hipMemcpyDeviceToHost);
    hipMemcpy(hPiR.data(), dPiR, hPiR.size() * 8, hipMemcpyDeviceToHost);
    for (int b = 0; b < nblocks; b++) {
        const int s0 = b * blen, e0 = std::min(s0 + blen, S);
        std::vector<cd> x((size_t)Np * C), y((size_t)Np * C);
        for (int r = 0; r < Np; r++) for (int c = 0; c < C; c++) {
            if (MODE == 0) x[r * C + c] = (r == c) ? 1.0 : 0.0;
            else if (MODE == 2) x[r * C + c] = 0.0;
            else x[r * C + c] = cd(start[(size_t)b * hstep + panel_at(r, c)], start[(size_t)b * hstep + panel_at(r, c) + 8]);
        }
        for (int st = 0; st < e0 - s0; st++) {
            const int n = (MODE >= 2) ? e0 - 1 - st : s0 + st;
            for (int r = 0; r < Np; r++) for (int c = 0; c < C; c++) {
                cd s = 0;
                for (int k = 0; k < Np; k++) s += ((MODE >= 2) ? std::conj(P[n * pl + k * Np + r]) : P[n * pl + r * Np + k]) * x[k * C + c];
                if (MODE >= 2) s += cd(forcing[(size_t)n * hstep + panel_at(r, c)], forcing[(size_t)n * hstep + panel_at(r, c) + 8]);
                y[r * C + c] = s;
            }
            x.swap(y);
            if (MODE == 1 || MODE == 3) {
                const int nout = (MODE == 3) ? n : n + 1;
                for (int r = 0; r < Np; r++) for (int c = 0; c < C; c++) {
                    const cd d(hout[(size_t)nout * hstep + panel_at(r, c)], hout[(size_t)nout * hstep + panel_at(r, c) + 8]);
                    err = fmax(err, std::abs(d - x[r * C + c])); ref_max = fmax(ref_max, std::abs(x[r * C + c]));
                }
            }
        }
        for (int r = 0; r < Np; r++) for (int c = 0; c < C; c++) {
            cd d;
            if (MODE == 0) d = cd(hPiR[(size_t)b * 2 * pl + (size_t)r * 2 * Np + (c >> 3) * 16 + (c & 7)], hPiR[(size_t)b * 2 * pl + (size_t)r * 2 * Np + (c >> 3) * 16 + 8 + (c & 7)]);
            else if (MODE == 2) d = cd(hphi[(size_t)b * hstep + panel_at(r, c)], hphi[(size_t)b * hstep + panel_at(r, c) + 8]);
            else continue;
            err = fmax(err, std::abs(d - x[r * C + c])); ref_max = fmax(ref_max, std::abs(x[r * C + c]));
        }
    }
    printf("max |device - host| = %.3e (max |ref| %.3e)\n", err, ref_max);
#ifdef QGD_CHAIN_PROFILE
    long long pr[64 * 8];
    hipMemcpyFromSymbol(pr, HIP_SYMBOL(g_chain_prof), sizeof pr);
    printf("step: wake->mfma_done  ->lds_written  ->barrier_passed  ->stores_issued  ->prefetch_issued | wake(st) - barrier_passed(st-1)   [cycles, block 0]\n");
    for (int st = 0; st < std::min(blen, 16); st++)
        printf("%3d: %6lld %6lld %6lld %6lld %6lld | %6lld\n", st, pr[st * 8 + 1] - pr[st * 8], pr[st * 8 + 2] - pr[st * 8 + 1], pr[st * 8 + 3] - pr[st * 8 + 2],
               pr[st * 8 + 4] - pr[st * 8 + 3], pr[st * 8 + 5] - pr[st * 8 + 4], st ? pr[st * 8] - pr[(st - 1) * 8 + 3] : 0LL);
#endif
    return 0;
}
