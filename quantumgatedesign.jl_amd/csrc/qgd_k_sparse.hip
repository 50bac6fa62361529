// qgd_k_sparse.hip -- sparse-operator path of the step matrices and of the gradient scalars.
// (conventions and layouts: qgd_kernels_common.h; algorithm: DESIGN.md)
//
// The operators of the reference's physical problems (multi_qudit_systems.jl: drift diagonal,
// controls a_k +/- a_k^dagger lifted by Kronecker products) have a handful of non-zeros per row;
// the dense MFMA kernels (qgd_k_build.hip, qgd_k_grad.hip) spend N/nnz times the necessary
// flops on them.  Here A_d(t_n) = sum_o coef * op_o is kept in ELL form over the UNION sparsity
// pattern of all operators (built once on the host, qgd_host_alloc.cpp): lane = matrix row, each
// thread owns a few complex columns, the neighbour rows of the current source are fetched from
// LDS (row stride padded so that 16 consecutive rows cover all 64 banks).  fp64 VALU, no MFMA:
// the flops left after exploiting the sparsity are below the cost of writing L and R.
#include "qgd_kernels_common.h"
#include "qgd_ell.h"


// -DQGD_SPARSE_PROFILE: cycles of workgroup 0 / thread 0 between the marks below, summed over launches
// (read back with qgdk_sparse_profile; development aid, not compiled into the shipped library)
#ifdef QGD_SPARSE_PROFILE
__device__ unsigned long long g_sparse_prof[32];
#define SP_PROF_BEGIN long long prof_last_ = clock64();
#define SP_PROF(i) do { if (blockIdx.x == 0 && blockIdx.y == 1 && threadIdx.x == 0) { const long long now_ = clock64(); atomicAdd(&g_sparse_prof[i], (unsigned long long)(now_ - prof_last_)); prof_last_ = now_; } } while (0)
#else
#define SP_PROF_BEGIN
#define SP_PROF(i) do { } while (0)
#endif
// ---------------------------------------------------------------------------
// K1 (sparse path, Np <= 64): L_n and R_n from the Taylor recursion on the identity,
//   D_{i+1} = 1/(i+1) sum_{s<=i} A_{i-s} D_s,  L = sum cL_j D_j,  R = sum cR_j D_j,
// one workgroup = (time point, slab of 32 complex columns), 8 waves x 4 columns, lane = row.
// Source-major like the dense kernel: when D_i is complete its contributions to every later
// level are accumulated at once, so D_i's neighbour rows are read from LDS once.
// ---------------------------------------------------------------------------
template <int M, int NW, int NOPS>
__global__ __launch_bounds__(64 * NW) void k_build_LR_ell(const int32_t *__restrict__ ell_col,
                                                      const uint8_t *__restrict__ ell_inv,
                                                      const double *__restrict__ ell_val,
                                                      const double *__restrict__ tab,
                                                      double *__restrict__ L, double *__restrict__ R,
                                                      const double *__restrict__ cw, int Np, int n_ops, int Z)
{
    constexpr int NTH = 64 * NW, CW = 4 * NW;           // threads, complex columns per workgroup (NW waves x 4 columns)
    constexpr int DS = CW + 1;                          // row stride of the source slab in complex numbers
    extern __shared__ __attribute__((aligned(16))) double smem_raw[];
    c2 *As = reinterpret_cast<c2 *>(smem_raw);          // [M][Z][64]
    c2 *Ds = As + (size_t)M * Z * 64;                   // [64][DS]   (also the output staging area)
    int *Ecol = reinterpret_cast<int *>(Ds + 64 * DS);  // [Z][64]
    const int n = blockIdx.x, slab = blockIdx.y;
    const int tid = threadIdx.x, w = tid >> 6, r = tid & 63;
    const int vc = min(CW, Np - slab * CW);             // valid columns of this slab
    const int cl = 4 * w, c0 = slab * CW + cl;          // first owned column (slab-local / global)
    const bool active = (r < Np) && (cl < vc);
    SP_PROF_BEGIN

    // (every global load of the prologue is issued before the first wait: the neighbour lists travel to registers while A_d(t_n)
    //  is assembled -- as loops of their own, load / wait / ds_write, they were a second round trip in each of the kernel's three
    //  rounds of workgroups)
    constexpr int EIT = 1024 / NTH;                     // Z <= 16: Z * 64 <= EIT * NTH
    int ecv[EIT];
    #pragma unroll
    for (int it = 0; it < EIT; it++) {
        const int item = tid + it * NTH, rr = item & 63, e = item >> 6;
        ecv[it] = (item < Z * 64 && rr < Np) ? ell_col[(size_t)e * Np + rr] : 0;
    }
    const uint32_t slots = active ? *reinterpret_cast<const uint32_t *>(ell_inv + (size_t)r * Np + c0) : 0xffffffffu;
    assemble_ell<NOPS>(As, ell_val, tab, n, M, M, n_ops, Z, Np, tid, NTH);
    #pragma unroll
    for (int it = 0; it < EIT; it++) {
        const int item = tid + it * NTH;
        if (item < Z * 64) Ecol[item] = ecv[it];
    }
    for (int item = tid + EIT * NTH; item < Z * 64; item += NTH) {      // (not reached for Z <= 16)
        const int rr = item & 63, e = item >> 6;
        Ecol[item] = (rr < Np) ? ell_col[(size_t)e * Np + rr] : 0;
    }
    SP_PROF(0);
    c2 T[M][4], Lacc[4], Racc[4];
    #pragma unroll
    for (int c = 0; c < 4; c++) {
        #pragma unroll
        for (int q = 0; q < M; q++) T[q][c] = (c2){0.0, 0.0};
        const double id = (r == c0 + c) ? 1.0 : 0.0;
        Lacc[c] = (c2){id, 0.0}; Racc[c] = (c2){id, 0.0};
    }
    __syncthreads();
    SP_PROF(1);
    // source 0 is the identity: A_d I = A_d.  ell_inv[r][col] = slot of column col in row r (0xff: none)
    if (active) {
        #pragma unroll
        for (int c = 0; c < 4; c++) {
            const int e = (slots >> (8 * c)) & 0xff;
            if (e != 0xff) {
                #pragma unroll
                for (int d = 0; d < M; d++) {
                    const c2 a = As[(d * Z + e) * 64 + r];
                    T[d][c].re += a.re; T[d][c].im += a.im;
                }
            }
        }
    }
    SP_PROF(2);
    #pragma unroll
    for (int i = 1; i <= M; i++) {
        const double inv = 1.0 / (double)i, cL = cw[2 * i + 1], cR = cw[2 * i];
        c2 Di[4];
        #pragma unroll
        for (int c = 0; c < 4; c++) {
            Di[c] = (c2){T[i - 1][c].re * inv, T[i - 1][c].im * inv};
            Lacc[c].re = __builtin_fma(cL, Di[c].re, Lacc[c].re); Lacc[c].im = __builtin_fma(cL, Di[c].im, Lacc[c].im);
            Racc[c].re = __builtin_fma(cR, Di[c].re, Racc[c].re); Racc[c].im = __builtin_fma(cR, Di[c].im, Racc[c].im);
        }
        if (i == M) break;
        // (a wave reads only the columns it wrote: the exchange is among its own lanes, no workgroup barrier)
        wave_lds_fence();                               // the previous source has been consumed
        #pragma unroll
        for (int c = 0; c < 4; c++) Ds[r * DS + cl + c] = Di[c];
        wave_lds_fence();
        if (active) {
            // (an explicit one-entry-ahead prefetch of the neighbour row and operator values was measured: 66.5 vs
            //  62.9 us -- the loop is not waiting on LDS latency)
            _Pragma("unroll 1") for (int e = 0; e < Z; e++) {
                const c2 *src = Ds + Ecol[e * 64 + r] * DS + cl;
                c2 x[4];
                #pragma unroll
                for (int c = 0; c < 4; c++) x[c] = src[c];
                #pragma unroll
                for (int d = 0; d + i < M; d++) {
                    const c2 a = As[(d * Z + e) * 64 + r];
                    #pragma unroll
                    for (int c = 0; c < 4; c++) cfma(T[i + d][c], a, x[c]);
                }
            }
        }
    }
    SP_PROF(3);
#ifdef QGD_BUILD_DIRECT_OUTPUT
    // (measured alternative: output straight from the registers -- a thread's 4 columns are 32 contiguous bytes of the
    //  real and of the imaginary half of a panel row, the other waves fill the rest of each 128-byte line.  No staging,
    //  four barriers fewer, and SLOWER: 46.7 against 40.4 us on the 551-point grid -- quarter-line writes.)
    if (active) {
        const int PW = 2 * Np;
        const size_t o = (size_t)n * Np * PW + (size_t)r * PW + (c0 >> 3) * 16 + (c0 & 7);
        typedef double dbl4 __attribute__((ext_vector_type(4)));
        *reinterpret_cast<dbl4 *>(L + o) = (dbl4){Lacc[0].re, Lacc[1].re, Lacc[2].re, Lacc[3].re};
        *reinterpret_cast<dbl4 *>(L + o + 8) = (dbl4){Lacc[0].im, Lacc[1].im, Lacc[2].im, Lacc[3].im};
        *reinterpret_cast<dbl4 *>(R + o) = (dbl4){Racc[0].re, Racc[1].re, Racc[2].re, Racc[3].re};
        *reinterpret_cast<dbl4 *>(R + o + 8) = (dbl4){Racc[0].im, Racc[1].im, Racc[2].im, Racc[3].im};
    }
    SP_PROF(4);
#else
    // output through LDS so that panel rows are written as contiguous segments
    const int PW = 2 * Np, SW = 2 * vc;                 // panel width, width of this slab's part of a row
    constexpr int SS = 2 * CW + 1;
    double *st = reinterpret_cast<double *>(Ds);        // [64][SS]
    const int row0 = tid / SW, k0 = tid % SW, rstep = NTH / SW, kstep = NTH % SW, nit = (Np * SW + NTH - 1) / NTH;
    #pragma unroll
    for (int pass = 0; pass < 2; pass++) {
        lds_barrier();                                  // (LDS-only: the stores of L stay in flight while R is staged)
        #pragma unroll
        for (int c = 0; c < 4; c++) {
            const c2 v = pass ? Racc[c] : Lacc[c];
            const int lc = cl + c, o = r * SS + (lc >> 3) * 16 + (lc & 7);
            st[o] = v.re; st[o + 8] = v.im;
        }
        lds_barrier();
        double *dst = (pass ? R : L) + (size_t)n * Np * PW + slab * 2 * CW;
        // item = tid, tid + NTH, ...: (row, k) = (item / SW, item % SW) advanced by (NTH / SW, NTH % SW) -- the division per item
        // was a third of the kernel's vector instructions -- and four LDS reads in flight ahead of their stores
        int row = row0, k = k0;
        for (int it0 = 0; it0 < nit; it0 += 4) {
            double v[4]; int at[4];
            #pragma unroll
            for (int u = 0; u < 4; u++) {
                const bool on = it0 + u < nit && row < Np;
                at[u] = on ? row * PW + k : -1;
                v[u] = on ? st[row * SS + k] : 0.0;
                k += kstep; row += rstep;
                if (k >= SW) { k -= SW; row++; }
            }
            #pragma unroll
            for (int u = 0; u < 4; u++)
                if (at[u] >= 0) dst[at[u]] = v[u];
        }
    }
    SP_PROF(4);
#endif
}

// ---------------------------------------------------------------------------
// K9+K10 (sparse path, Np <= 64): everything the gradient needs at one time point (the passes
// of the dense k_gradpoint64, qgd_k_grad.hip, reordered so that one state slab in LDS is enough).
// Workgroup = (time point, group of 8 columns), 4 waves x 2 columns, lane = row.
//   G passes first (they depend on lambda only): g_i += -(1/j) A_{j-1-i} g_j, g_1..g_m in registers;
//   then source-major over psi_i, i = 0..m-1 (only the current source is in LDS):
//     D part  T[i+d+1] += A_d psi_i,   psi_{i+1} = T[i+1]/(i+1)
//     S part  per operator o (its own ELL list): U = Sym_o psi_i, V = Asym_o psi_i,
//             sigP[o][j-1-i] += (1/j) Re<-iU, g_j>,  sigQ[o][j-1-i] += (1/j) Re<V, g_j>,  j > i.
// ---------------------------------------------------------------------------
template <int M, int NOPS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(M <= 4 ? 3 : 2, M <= 4 ? 3 : 2)))
void k_gradpoint_ell(const int32_t *__restrict__ ell_col,
                                                       const double *__restrict__ ell_val,
                                                       const int32_t *__restrict__ op_col,
                                                       const double *__restrict__ op_val,
                                                       const double *__restrict__ tab,
                                                       const double *__restrict__ hist,
                                                       const double *__restrict__ lam,
                                                       double *__restrict__ sigma,
                                                       const double *__restrict__ cw, int Np, int cp,
                                                       int nt, int n_ops, int Z, int Zo,
                                                       const double *__restrict__ G, const int64_t *__restrict__ goff,
                                                       const int32_t *__restrict__ ncoef, const int32_t *__restrict__ poff,
                                                       double *__restrict__ cpart, int n_pcof, int g_nt, int g_n0)
{
    constexpr int PS = 9, ND = (M > 1) ? M - 1 : 1, NO = NOPS_LIM(NOPS);
    extern __shared__ __attribute__((aligned(16))) double smem_raw[];
    c2 *As = reinterpret_cast<c2 *>(smem_raw);          // [ND][Z][64]
    c2 *buf = As + (size_t)ND * Z * 64;                 // [64][PS]   the current right operand
    c2 *Ov = buf + 64 * PS;                             // [n_ops][Zo][64] (Asym_o, Sym_o) values
    double *red = reinterpret_cast<double *>(Ov + (size_t)n_ops * Zo * 64);   // [n_ops*M*2][16] partial sums
    int *Ecol = reinterpret_cast<int *>(red + n_ops * M * 2 * 16);      // [Z][64]
    int *Ocol = Ecol + Z * 64;                          // [n_ops][Zo][64]
    const int grp = blockIdx.x, n = blockIdx.y;
    const int tid = threadIdx.x, w = tid >> 6, r = tid & 63, cl = 2 * w;
    const int PWc = 2 * cp;
    const size_t hstep = (size_t)Np * PWc;
    const bool live = r < Np;
    SP_PROF_BEGIN

    // (global loads first, in the order of their latency: the seeds and psi_0 go to registers and are in flight
    //  while the lists and A_d(t_n) are brought into LDS)
    // own elements of psi_0 and the seeds g_j = c_j dt^j lam_{n+1} - c_j (-dt)^j lam_n
    c2 ps[2], g[M][2];
    #pragma unroll
    for (int c = 0; c < 2; c++) {
        const size_t o = (size_t)(live ? r : 0) * PWc + grp * 16 + cl + c;
        ps[c] = live ? (c2){hist[(size_t)n * hstep + o], hist[(size_t)n * hstep + o + 8]} : (c2){0.0, 0.0};
        const c2 ln = (live && n >= 1) ? (c2){lam[(size_t)n * hstep + o], lam[(size_t)n * hstep + o + 8]} : (c2){0.0, 0.0};
        const c2 lx = (live && n <= nt - 2) ? (c2){lam[(size_t)(n + 1) * hstep + o], lam[(size_t)(n + 1) * hstep + o + 8]} : (c2){0.0, 0.0};
        #pragma unroll
        for (int j = 1; j <= M; j++)
            g[j - 1][c] = (c2){cw[2 * j] * lx.re - cw[2 * j + 1] * ln.re, cw[2 * j] * lx.im - cw[2 * j + 1] * ln.im};
    }
    // the neighbour lists and the operator values: loaded to registers here, written to LDS behind the assembly (one round trip
    // for the whole prologue instead of one per list)
    constexpr int EIT = 4, OIT = 2;                     // Z <= 16: Z * 64 <= EIT * 256; the first OIT * 256 operator entries
    int ecv[EIT], ocv[OIT];
    c2 ovv[OIT];
    #pragma unroll
    for (int it = 0; it < EIT; it++) {
        const int item = tid + it * 256, rr = item & 63, e = item >> 6;
        ecv[it] = (item < Z * 64 && rr < Np) ? ell_col[(size_t)e * Np + rr] : 0;
    }
    #pragma unroll
    for (int it = 0; it < OIT; it++) {
        const int item = tid + it * 256, rr = item & 63, oe = item >> 6, o = oe / Zo, e = oe % Zo;
        const bool ok = item < n_ops * Zo * 64 && rr < Np;
        ocv[it] = ok ? op_col[(size_t)oe * Np + rr] : 0;
        ovv[it] = ok ? (c2){op_val[((size_t)(2 * o) * Zo + e) * Np + rr], op_val[((size_t)(2 * o + 1) * Zo + e) * Np + rr]} : (c2){0.0, 0.0};
    }
    if (M > 1) assemble_ell<NOPS>(As, ell_val, tab, n, M, ND, n_ops, Z, Np, tid, 256);
    SP_PROF(16);
    #pragma unroll
    for (int it = 0; it < EIT; it++) {
        const int item = tid + it * 256;
        if (item < Z * 64) Ecol[item] = ecv[it];
    }
    for (int item = tid + EIT * 256; item < Z * 64; item += 256) {      // (not reached for Z <= 16)
        const int rr = item & 63, e = item >> 6;
        Ecol[item] = (rr < Np) ? ell_col[(size_t)e * Np + rr] : 0;
    }
    #pragma unroll
    for (int it = 0; it < OIT; it++) {
        const int item = tid + it * 256;
        if (item < n_ops * Zo * 64) { Ocol[item] = ocv[it]; Ov[item] = ovv[it]; }
    }
    for (int item = tid + OIT * 256; item < n_ops * Zo * 64; item += 256) {
        const int rr = item & 63, oe = item >> 6, o = oe / Zo, e = oe % Zo;
        const bool ok = rr < Np;
        Ocol[item] = ok ? op_col[(size_t)oe * Np + rr] : 0;
        Ov[item] = ok ? (c2){op_val[((size_t)(2 * o) * Zo + e) * Np + rr], op_val[((size_t)(2 * o + 1) * Zo + e) * Np + rr]} : (c2){0.0, 0.0};
    }
    SP_PROF(17);
    // ---- G passes: g_i += -(1/j) A_{j-1-i} g_j (A^H = -A), i = 1..j-1, j = m..2
    __syncthreads();                                    // the lists and A_d(t_n) are in LDS
    // (from here on a wave reads only the two columns of `buf` it wrote itself: the exchange is among its own lanes)
    #pragma unroll
    for (int j = M; j >= 2; j--) {
        wave_lds_fence();                               // the previous right operand has been consumed
        buf[r * PS + cl] = g[j - 1][0]; buf[r * PS + cl + 1] = g[j - 1][1];
        wave_lds_fence();
        c2 t[M][2];
        #pragma unroll
        for (int q = 0; q < M; q++) { t[q][0] = (c2){0.0, 0.0}; t[q][1] = (c2){0.0, 0.0}; }
        _Pragma("unroll 1") for (int e = 0; e < Z; e++) {
            const c2 *src = buf + (size_t)Ecol[e * 64 + r] * PS + cl;
            const c2 x0 = src[0], x1 = src[1];
            #pragma unroll
            for (int i = 1; i <= j - 1; i++) {
                const c2 a = As[((j - 1 - i) * Z + e) * 64 + r];
                cfma(t[i][0], a, x0); cfma(t[i][1], a, x1);
            }
        }
        const double sc = -1.0 / (double)j;
        #pragma unroll
        for (int i = 1; i <= j - 1; i++)
            #pragma unroll
            for (int c = 0; c < 2; c++) {
                g[i - 1][c].re = __builtin_fma(sc, t[i][c].re, g[i - 1][c].re);
                g[i - 1][c].im = __builtin_fma(sc, t[i][c].im, g[i - 1][c].im);
            }
    }
    SP_PROF(18);
    // ---- sources psi_0..psi_{m-1}
    c2 T[M][2];
    double sp[NO][M], sq[NO][M];
    #pragma unroll
    for (int q = 0; q < M; q++) { T[q][0] = (c2){0.0, 0.0}; T[q][1] = (c2){0.0, 0.0}; }
    #pragma unroll
    for (int o = 0; o < NO; o++)
        #pragma unroll
        for (int d = 0; d < M; d++) { sp[o][d] = 0.0; sq[o][d] = 0.0; }
    #pragma unroll
    for (int i = 0; i < M; i++) {
        wave_lds_fence();
        buf[r * PS + cl] = ps[0]; buf[r * PS + cl + 1] = ps[1];
        wave_lds_fence();
        if (i + 1 < M) {
            _Pragma("unroll 1") for (int e = 0; e < Z; e++) {
                const c2 *src = buf + (size_t)Ecol[e * 64 + r] * PS + cl;
                const c2 x0 = src[0], x1 = src[1];
                #pragma unroll
                for (int d = 0; d + i + 1 < M; d++) {
                    const c2 a = As[(d * Z + e) * 64 + r];
                    cfma(T[i + d + 1][0], a, x0); cfma(T[i + d + 1][1], a, x1);
                }
            }
        }
        SP_PROF(19);
        #pragma unroll
        for (int o = 0; o < NO; o++) {
            if (!NOPS_ON(NOPS, o, n_ops)) continue;
            c2 U[2] = {{0.0, 0.0}, {0.0, 0.0}}, V[2] = {{0.0, 0.0}, {0.0, 0.0}};
            _Pragma("unroll 1") for (int e = 0; e < Zo; e++) {
                const int at = (o * Zo + e) * 64 + r;
                const c2 kv = Ov[at];                   // (Asym_o, Sym_o)
                const c2 *src = buf + (size_t)Ocol[at] * PS + cl;
                #pragma unroll
                for (int c = 0; c < 2; c++) {
                    const c2 x = src[c];
                    U[c].re = __builtin_fma(kv.im, x.re, U[c].re); U[c].im = __builtin_fma(kv.im, x.im, U[c].im);
                    V[c].re = __builtin_fma(kv.re, x.re, V[c].re); V[c].im = __builtin_fma(kv.re, x.im, V[c].im);
                }
            }
            #pragma unroll
            for (int j = i + 1; j <= M; j++) {
                double ap = 0.0, aq = 0.0;
                #pragma unroll
                for (int c = 0; c < 2; c++) {
                    ap += U[c].im * g[j - 1][c].re - U[c].re * g[j - 1][c].im;
                    aq += V[c].re * g[j - 1][c].re + V[c].im * g[j - 1][c].im;
                }
                sp[o][j - 1 - i] += ap * (1.0 / (double)j);
                sq[o][j - 1 - i] += aq * (1.0 / (double)j);
                // keep the running sums materialised here (the unrolled code otherwise keeps every
                // product alive until the final reduction: 2x the register count)
                asm volatile("" : "+v"(sp[o][j - 1 - i]), "+v"(sq[o][j - 1 - i]));
            }
        }
        SP_PROF(20);
        if (i + 1 < M) {
            const double inv = 1.0 / (double)(i + 1);
            ps[0] = (c2){T[i + 1][0].re * inv, T[i + 1][0].im * inv};
            ps[1] = (c2){T[i + 1][1].re * inv, T[i + 1][1].im * inv};
        }
    }
    SP_PROF(21);
    #pragma unroll
    for (int o = 0; o < NO; o++) {
        if (!NOPS_ON(NOPS, o, n_ops)) continue;
        #pragma unroll
        for (int d = 0; d < M; d++) {
            const double a = row16_sum(live ? sp[o][d] : 0.0), b = row16_sum(live ? sq[o][d] : 0.0);
            if ((r & 15) == 15) {                       // one partial sum per 16 rows and wave
                red[((o * M + d) * 2) * 16 + (tid >> 4)] = a;
                red[((o * M + d) * 2 + 1) * 16 + (tid >> 4)] = b;
            }
        }
    }
    __syncthreads();
    double sv[(NO * M * 2 + 255) / 256];
    #pragma unroll
    for (int it = 0; it < (NO * M * 2 + 255) / 256; it++) {
        const int e = tid + it * 256;
        double v = 0.0;
        if (e < n_ops * M * 2) {
            #pragma unroll
            for (int q = 0; q < 16; q++) v += red[e * 16 + q];
            sigma[((size_t)grp * nt + n) * n_ops * M * 2 + e] = v;      // this column group's plane: stored, not accumulated
        }
        sv[it] = v;
    }
    // Contraction with the control basis for THIS time point (the "grad_slice .-= contrib" of
    // eval_grad_discrete_adjoint.jl:642-643): row (column group, n) of cpart, which k_contract_sum adds over the rows in
    // row order -- no atomics between here and grad, so the gradient is bitwise reproducible.  14 KB of G per workgroup.
    __syncthreads();                                    // every partial sum has been read: red is free
    #pragma unroll
    for (int it = 0; it < (NO * M * 2 + 255) / 256; it++) { const int e = tid + it * 256; if (e < n_ops * M * 2) red[e] = sv[it]; }
    __syncthreads();
    for (int p = tid; p < n_pcof; p += 256) {
        int kq = 0;
        while (kq + 1 < n_ops && p >= poff[kq + 1]) kq++;
        const int l = p - poff[kq], nc = ncoef[kq];
        const double *gp = G + goff[kq] + ((size_t)(n + g_n0) * (M + 1)) * nc + l;
        const double *gq = gp + (size_t)g_nt * (M + 1) * nc;
        double acc = 0.0;
        #pragma unroll
        for (int d = 0; d < M; d++) acc += gp[(size_t)d * nc] * red[(kq * M + d) * 2] + gq[(size_t)d * nc] * red[(kq * M + d) * 2 + 1];
        cpart[((size_t)grp * nt + n) * n_pcof + p] = -acc;
    }
    SP_PROF(22);
}

// ---------------------------------------------------------------------------
// Stage derivatives w_1..w_m of the stored states (compute_derivatives!, hermite.jl:56-101) for the uv_history output:
//   (j+1) w_{j+1} = sum_{i<=j} A_{j-i} w_i,  A_d(t_n) assembled once per workgroup into LDS (ELL), lane = row,
// two columns per thread -- the "sources" half of k_gradpoint_ell, writing the panels dpsi[n][j-1] instead of inner
// products.  (The MFMA kernel k_derivs took 141 us on the cnot3 grid for what is 7 entries per row.)
// ---------------------------------------------------------------------------
template <int M, int NOPS>
__global__ __launch_bounds__(256) void k_derivs_ell(const int32_t *__restrict__ ell_col, const double *__restrict__ ell_val,
                                                    const double *__restrict__ tab, const double *__restrict__ hist,
                                                    double *__restrict__ dpsi, int Np, int cp, int n_ops, int Z)
{
    constexpr int PS = 9;
    extern __shared__ __attribute__((aligned(16))) double smem_raw[];
    c2 *As = reinterpret_cast<c2 *>(smem_raw);          // [M][Z][64]
    c2 *buf = As + (size_t)M * Z * 64;                  // [64][PS]   the current source
    int *Ecol = reinterpret_cast<int *>(buf + 64 * PS); // [Z][64]
    const int grp = blockIdx.x, n = blockIdx.y;
    const int tid = threadIdx.x, w = tid >> 6, r = tid & 63, cl = 2 * w;
    const int PWc = 2 * cp;
    const size_t hstep = (size_t)Np * PWc;
    const bool live = r < Np;
    const size_t o0 = (size_t)(live ? r : 0) * PWc + grp * 16 + cl;
    c2 ps[2];
    #pragma unroll
    for (int c = 0; c < 2; c++)
        ps[c] = live ? (c2){hist[(size_t)n * hstep + o0 + c], hist[(size_t)n * hstep + o0 + c + 8]} : (c2){0.0, 0.0};
    assemble_ell<NOPS>(As, ell_val, tab, n, M, M, n_ops, Z, Np, tid, 256);
    for (int item = tid; item < Z * 64; item += 256) {
        const int rr = item & 63, e = item >> 6;
        Ecol[item] = (rr < Np) ? ell_col[(size_t)e * Np + rr] : 0;
    }
    c2 T[M + 1][2];
    #pragma unroll
    for (int q = 0; q <= M; q++) { T[q][0] = (c2){0.0, 0.0}; T[q][1] = (c2){0.0, 0.0}; }
    __syncthreads();                                    // the list and A_d(t_n) are in LDS
    #pragma unroll
    for (int i = 0; i < M; i++) {
        wave_lds_fence();                               // (a wave reads only the two columns it wrote)
        buf[r * PS + cl] = ps[0]; buf[r * PS + cl + 1] = ps[1];
        wave_lds_fence();
        _Pragma("unroll 1") for (int e = 0; e < Z; e++) {
            const c2 *src = buf + (size_t)Ecol[e * 64 + r] * PS + cl;
            const c2 x0 = src[0], x1 = src[1];
            #pragma unroll
            for (int d = 0; d + i + 1 <= M; d++) {
                const c2 a = As[(d * Z + e) * 64 + r];
                cfma(T[i + d + 1][0], a, x0); cfma(T[i + d + 1][1], a, x1);
            }
        }
        const double inv = 1.0 / (double)(i + 1);
        ps[0] = (c2){T[i + 1][0].re * inv, T[i + 1][0].im * inv};
        ps[1] = (c2){T[i + 1][1].re * inv, T[i + 1][1].im * inv};
        if (live) {
            double *dst = dpsi + ((size_t)n * M + i) * hstep + o0;
            dst[0] = ps[0].re; dst[1] = ps[1].re; dst[8] = ps[0].im; dst[9] = ps[1].im;
        }
    }
}
static size_t lds_derivs_ell(int M, int Z) { return ((size_t)M * Z * 64 + 64 * 9) * 16 + (size_t)Z * 64 * 4; }

static size_t lds_build_ell(int M, int Z, int NW) { return ((size_t)M * Z * 64 + 64 * (4 * NW + 1)) * 16 + (size_t)Z * 64 * 4; }
static size_t lds_grad_ell(int M, int Z, int n_ops, int Zo)
{
    const int nd = (M > 1) ? M - 1 : 1;
    return ((size_t)nd * Z * 64 + 64 * 9 + (size_t)n_ops * Zo * 64) * 16 + (size_t)n_ops * M * 2 * 16 * 8 +
           ((size_t)Z * 64 + (size_t)n_ops * Zo * 64) * 4;
}

// Workgroup = (time point, 4*NW columns), NW waves.  NW = 8 (32 columns, 64 KB of LDS, two workgroups per CU) is the
// only instantiation: 46.5 us on the 551-point benchmark grid against 49.8 us for NW = 4 (16 columns, 48 KB, three per CU,
// measured in round 2) -- the narrow unit packs the CUs better but assembles A_d(t_n) four times per time point.  (Round 5, with
// today's kernel: 40.2-40.4 us for NW = 4 against 39.4-40.3.)
// (Before the LDS accesses were 16-byte aligned both took 63 us, bound by bank conflicts.)
template <int M, int NW, int NOPS>
static int launch_build_ell_nw(const qgdk_ctx *c)
{
    const size_t shm = lds_build_ell(M, c->ell_z, NW);
    SET_LDS_ONCE((k_build_LR_ell<M, NW, NOPS>), shm);
    hipLaunchKernelGGL((k_build_LR_ell<M, NW, NOPS>), dim3(c->nt, (c->Np + 4 * NW - 1) / (4 * NW)), dim3(64 * NW), shm, c->stream,
                       c->ell_col, c->ell_inv, c->ell_val, c->tab, c->L, c->R, c->cw, c->Np, c->n_ops, c->ell_z);
    return (int)hipGetLastError();
}

template <int M>
static int launch_build_ell(const qgdk_ctx *c)
{
#define CALL_BE(N) return launch_build_ell_nw<M, 8, N>(c)
    DISPATCH_NOPS(c->n_ops, CALL_BE)
#undef CALL_BE
    return 0;
}

template <int M, int NOPS>
static int launch_grad_ell_n(const qgdk_ctx *c)
{
    const size_t shm = lds_grad_ell(M, c->ell_z, c->n_ops, c->op_z);
    SET_LDS_ONCE((k_gradpoint_ell<M, NOPS>), shm);
    hipLaunchKernelGGL((k_gradpoint_ell<M, NOPS>), dim3(c->cp / 8, c->nt), dim3(256), shm, c->stream, c->ell_col, c->ell_val,
                       c->op_col, c->op_val, c->tab, c->hist, c->lam, c->sigma, c->cw, c->Np, c->cp, c->nt, c->n_ops,
                       c->ell_z, c->op_z, c->G, c->goff, c->ncoef, c->poff, c->cpart, c->n_pcof, c->g_nt ? c->g_nt : c->nt, c->g_n0);
    return (int)hipGetLastError();
}

template <int M>
static int launch_grad_ell(const qgdk_ctx *c)
{
#define CALL_GE(N) return launch_grad_ell_n<M, N>(c)
    DISPATCH_NOPS(c->n_ops, CALL_GE)
#undef CALL_GE
    return 0;
}

template <int M, int NOPS>
static int launch_derivs_ell_n(const qgdk_ctx *c)
{
    const size_t shm = lds_derivs_ell(M, c->ell_z);
    SET_LDS_ONCE((k_derivs_ell<M, NOPS>), shm);
    hipLaunchKernelGGL((k_derivs_ell<M, NOPS>), dim3(c->cp / 8, c->nt), dim3(256), shm, c->stream, c->ell_col, c->ell_val, c->tab,
                       c->hist, c->dpsi, c->Np, c->cp, c->n_ops, c->ell_z);
    return (int)hipGetLastError();
}

template <int M>
static int launch_derivs_ell(const qgdk_ctx *c)
{
#define CALL_DE(N) return launch_derivs_ell_n<M, N>(c)
    DISPATCH_NOPS(c->n_ops, CALL_DE)
#undef CALL_DE
    return 0;
}

#define DISPATCH_M(m, FN) \
    switch (m) { case 1: return FN<1>(c); case 2: return FN<2>(c); case 3: return FN<3>(c); case 4: return FN<4>(c); \
                 case 5: return FN<5>(c); case 6: return FN<6>(c); case 7: return FN<7>(c); case 8: return FN<8>(c); \
                 default: return (int)hipErrorInvalidValue; }

extern "C" {

// whether the sparse kernels can run this problem (pattern width Z, order 2m): LDS budget only
int qgdk_sparse_supported(int Np, int m, int n_ops, int Z)
{
    if (Np > 64 || m < 1 || m > 8 || Z < 1) return 0;
    return lds_build_ell(m, Z, 8) <= 150 * 1024 && lds_grad_ell(m, Z, n_ops, Z) <= 150 * 1024;
}

#ifdef QGD_SPARSE_PROFILE
int qgdk_sparse_profile(unsigned long long *out32, int reset)
{
    hipError_t e = hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_sparse_prof), 32 * sizeof(unsigned long long));
    if (reset) { unsigned long long z[32] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_sparse_prof), z, sizeof z); }
    return (int)e;
}
#endif

int qgdk_build_LR_sparse(const qgdk_ctx *c) { DISPATCH_M(c->m, launch_build_ell) }
int qgdk_gradient_sparse(const qgdk_ctx *c) { DISPATCH_M(c->m, launch_grad_ell) }
int qgdk_derivs_sparse(const qgdk_ctx *c) { DISPATCH_M(c->m, launch_derivs_ell) }

} // extern "C"
