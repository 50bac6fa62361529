// qgd_k_forced.hip -- forward-sensitivity ("forced") gradient, SURVEY.md section 8 row f2:
// the reference's own cross-check of the discrete adjoint (src/eval_grad_forced.jl:17-194, one forced
// forward sweep per control parameter).  (conventions and layouts: qgd_kernels_common.h)
//
// dA_d(t_n)/dtheta_l = Gp_l^(d)(t_n) (-i Sym_k) + Gq_l^(d)(t_n) Asym_k  (the controls are linear in
// pcof), so the forcing of every parameter of control k is a combination of 2m BASIS RESPONSES
// per time point: for (tau, d), with Omega = -i Sym_k (tau = p) or Asym_k (tau = q),
//   U_0 = 0,  U_{j+1} = ( sum_{i<=j} A_{j-i} U_i + [j >= d] Omega w_{j-d} ) / (j+1),
//   rhoR = sum_j c_j dt^j U_j,  rhoL = sum_j c_j (-dt)^j U_j,
// and the sensitivity of parameter l obeys
//   s_{n+1} = P_n s_n + L_{n+1}^-1 ( sum G_l(t_n) rhoR_n - sum G_l(t_{n+1}) rhoL_{n+1} ),  s_0 = 0.
// k_forced_basis forms rhoR, rhoL for all (k, tau, d) and n; k_forced_solve multiplies them by the
// L^-1 they meet; the sensitivities are propagated by the blocked scan of qgd_k_chain.hip
// (k_chain_fast MODE 4/5) with the block propagators of the forward sweep.
#include "qgd_kernels_common.h"

// acc(rows of block rb) = A * B for one 16-row block; A fragments from `afrag(arow, k, are, aim)`
template <class AFrag>
__device__ __forceinline__ d4 panel_product(int Np, int rb, int c16, int kk, const double *Bp, int ldb, AFrag afrag)
{
    d4 acc = (d4){0, 0, 0, 0};
    const int arow = rb * 16 + c16;
    for (int k0 = 0; k0 < Np; k0 += 4) {
        double are, aim, b1, b2;
        afrag(arow, k0 + kk, are, aim);
        panel_b(Bp + (size_t)(k0 + kk) * ldb, c16, b1, b2);
        acc = MFMA(are, b1, acc);
        acc = MFMA(aim, b2, acc);
    }
    return acc;
}

// One workgroup = (time point, state column group).  w_0 = hist[n], w_j = dpsi[n][j-1] (k_derivs).
// LDS: V[m] (Omega w_i), U[m] (U_1..U_m of the current basis), rhoR, rhoL: 2m+2 panels.
template <int NOPS>
__global__ __launch_bounds__(256) void k_forced_basis(const double *__restrict__ ops,
                                                      const double *__restrict__ tab,
                                                      const double *__restrict__ hist,
                                                      const double *__restrict__ dpsi,
                                                      double *__restrict__ BR, double *__restrict__ BL,
                                                      const double *__restrict__ cw, int Np, int cp,
                                                      int n_ops, int m, double *__restrict__ gscratch)
{
    extern __shared__ double lds_fb[];
    const int ps = Np * 16;
    // the 2m+2 work panels: LDS, or (N > 64 at high order: more than 150 KB) this workgroup's slab of an HBM scratch
    double *smem = gscratch ? gscratch + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (size_t)(2 * m + 2) * ps : lds_fb;
    double *V = smem, *U = smem + (size_t)m * ps, *rR = U + (size_t)m * ps, *rL = rR + ps;
    const int grp = blockIdx.x, n = blockIdx.y;
    const int PWc = 2 * cp;
    const size_t hstep = (size_t)Np * PWc, pl = (size_t)Np * Np;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c16 = lane & 15, kk = lane >> 4;
    const int NB = n_ops * 2 * m;
    auto wpanel = [&](int j) {      // global panel of w_j restricted to this column group (row stride PWc)
        return (j == 0 ? hist + (size_t)n * hstep : dpsi + ((size_t)n * m + (j - 1)) * hstep) + grp * 16;
    };
    for (int k = 0; k < n_ops; k++)
        for (int tau = 0; tau < 2; tau++) {
            // V_i = Omega w_i, Omega = -i Sym_k (K = 0, S = Sym_k) or Asym_k (K = Asym_k, S = 0)
            const double *plane = ops + (size_t)(tau == 0 ? 3 + 2 * k : 2 + 2 * k) * pl;
            for (int i = 0; i < m; i++)
                for (int rb = wave; rb * 16 < Np; rb += 4) {
                    const d4 acc = panel_product(Np, rb, c16, kk, wpanel(i), PWc,
                        [&](int arow, int kcol, double &are, double &aim) {
                            const double v = plane[(size_t)arow + (size_t)Np * kcol];
                            are = tau ? v : 0.0; aim = tau ? 0.0 : -v;
                        });
                    #pragma unroll
                    for (int r = 0; r < 4; r++) V[(size_t)i * ps + (rb * 16 + kk + 4 * r) * 16 + c16] = acc[r];
                }
            __threadfence_block(); __syncthreads();
            for (int d = 0; d < m; d++) {
                for (int e = threadIdx.x; e < ps; e += 256) { rR[e] = 0.0; rL[e] = 0.0; }
                __threadfence_block(); __syncthreads();
                for (int j = d; j < m; j++) {       // U_{j+1} = (V_{j-d} + sum_{i=d+1..j} A_{j-i} U_i)/(j+1), slot j
                    for (int rb = wave; rb * 16 < Np; rb += 4) {
                        d4 acc;
                        #pragma unroll
                        for (int r = 0; r < 4; r++) acc[r] = V[(size_t)(j - d) * ps + (rb * 16 + kk + 4 * r) * 16 + c16];
                        for (int i = d + 1; i <= j; i++) {
                            OpCoef cf;
                            load_coef(cf, tab, n, j - i, m, n_ops);
                            const d4 t = panel_product(Np, rb, c16, kk, U + (size_t)(i - 1) * ps, 16,
                                [&](int arow, int kcol, double &are, double &aim) {
                                    assembled_a<NOPS>(ops, Np, n_ops, cf, arow, kcol, are, aim);
                                });
                            #pragma unroll
                            for (int r = 0; r < 4; r++) acc[r] += t[r];
                        }
                        const double inv = 1.0 / (double)(j + 1), cR = cw[2 * (j + 1)], cL = cw[2 * (j + 1) + 1];
                        #pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const int o = (rb * 16 + kk + 4 * r) * 16 + c16;
                            const double u = acc[r] * inv;
                            U[(size_t)j * ps + o] = u;
                            rR[o] += cR * u; rL[o] += cL * u;
                        }
                    }
                    __threadfence_block(); __syncthreads();
                }
                const int b = (k * 2 + tau) * m + d;
                double *oR = BR + ((size_t)n * NB + b) * hstep + grp * 16, *oL = BL + ((size_t)n * NB + b) * hstep + grp * 16;
                for (int e = threadIdx.x; e < ps; e += 256) {
                    oR[(size_t)(e >> 4) * PWc + (e & 15)] = rR[e];
                    oL[(size_t)(e >> 4) * PWc + (e & 15)] = rL[e];
                }
                __threadfence_block(); __syncthreads();
            }
        }
}

// X[n][b] <- L_{n'}^-1 X[n][b] in place; side 0: BR with n' = n+1 (n < nt-1), side 1: BL with n' = n (n >= 1).
// LinvT: row-major planes.  One workgroup = (basis x column group, time point, side).
__global__ __launch_bounds__(256) void k_forced_solve(const double *__restrict__ LinvT,
                                                      double *__restrict__ BR, double *__restrict__ BL,
                                                      int Np, int cp, int NB, int nt)
{
    extern __shared__ double smem[];
    const int gpc = cp / 8, b = blockIdx.x / gpc, grp = blockIdx.x % gpc, n = blockIdx.y, side = blockIdx.z;
    const int nl = side ? n : n + 1;
    if (nl < 1 || nl > nt - 1) return;
    const int PWc = 2 * cp;
    const size_t hstep = (size_t)Np * PWc, pl = (size_t)Np * Np;
    double *X = (side ? BL : BR) + ((size_t)n * NB + b) * hstep + grp * 16;
    const double *T = LinvT + (size_t)nl * 2 * pl;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, c16 = lane & 15, kk = lane >> 4;
    for (int e = threadIdx.x; e < Np * 16; e += 256) smem[e] = X[(size_t)(e >> 4) * PWc + (e & 15)];
    __syncthreads();
    for (int rb = wave; rb * 16 < Np; rb += 4) {
        const d4 acc = panel_product(Np, rb, c16, kk, smem, 16,
            [&](int arow, int kcol, double &are, double &aim) {
                are = T[(size_t)arow * Np + kcol]; aim = T[pl + (size_t)arow * Np + kcol];
            });
        #pragma unroll
        for (int r = 0; r < 4; r++) X[(size_t)(rb * 16 + kk + 4 * r) * PWc + c16] = acc[r];
    }
}

// ---------------------------------------------------------------------------
// eval_forward with a user forcing (forward_evolution.jl:118-129,167-206): w' = A w + F with the scaled
// Taylor coefficients F_j(t_n) = F^(j)(t_n)/j! given.  compute_derivatives! adds F_j inside the
// recursion (hermite.jl:88-92): w_j = D_j w_0 + E_j with E_0 = 0, E_{j+1} = (sum_{i>=1} A_{j-i} E_i + F_j)/(j+1).
// This kernel forms E_1..E_m (kept for the derivative history) and rhoR = sum c_j dt^j E_j,
// rhoL = sum c_j (-dt)^j E_j; the step becomes psi_{n+1} = P_n psi_n + L_{n+1}^-1 (rhoR_n - rhoL_{n+1}).
// One workgroup = (time point, column group); LDS: E[m], rhoR, rhoL.
// ---------------------------------------------------------------------------
template <int NOPS>
__global__ __launch_bounds__(256) void k_forcing_terms(const double *__restrict__ ops,
                                                       const double *__restrict__ tab,
                                                       const double *__restrict__ F, double *__restrict__ E,
                                                       double *__restrict__ XR, double *__restrict__ XL,
                                                       const double *__restrict__ cw, int Np, int cp,
                                                       int n_ops, int m, double *__restrict__ gscratch)
{
    extern __shared__ double lds_ft[];
    const int ps = Np * 16;
    double *smem = gscratch ? gscratch + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (size_t)(m + 2) * ps : lds_ft;
    double *U = smem, *rR = U + (size_t)m * ps, *rL = rR + ps;
    const int grp = blockIdx.x, n = blockIdx.y;
    const int PWc = 2 * cp;
    const size_t hstep = (size_t)Np * PWc;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c16 = lane & 15, kk = lane >> 4;
    for (int e = threadIdx.x; e < ps; e += 256) { rR[e] = 0.0; rL[e] = 0.0; }
    __threadfence_block(); __syncthreads();
    for (int j = 0; j < m; j++) {
        const double *Fj = F + ((size_t)n * m + j) * hstep + grp * 16;
        for (int rb = wave; rb * 16 < Np; rb += 4) {
            d4 acc;
            #pragma unroll
            for (int r = 0; r < 4; r++) acc[r] = Fj[(size_t)(rb * 16 + kk + 4 * r) * PWc + c16];
            for (int i = 1; i <= j; i++) {
                OpCoef cf;
                load_coef(cf, tab, n, j - i, m, n_ops);
                const d4 t = panel_product(Np, rb, c16, kk, U + (size_t)(i - 1) * ps, 16,
                    [&](int arow, int kcol, double &are, double &aim) {
                        assembled_a<NOPS>(ops, Np, n_ops, cf, arow, kcol, are, aim);
                    });
                #pragma unroll
                for (int r = 0; r < 4; r++) acc[r] += t[r];
            }
            const double inv = 1.0 / (double)(j + 1), cR = cw[2 * (j + 1)], cL = cw[2 * (j + 1) + 1];
            #pragma unroll
            for (int r = 0; r < 4; r++) {
                const int o = (rb * 16 + kk + 4 * r) * 16 + c16;
                const double u = acc[r] * inv;
                U[(size_t)j * ps + o] = u;
                E[((size_t)n * m + j) * hstep + (size_t)(rb * 16 + kk + 4 * r) * PWc + grp * 16 + c16] = u;
                rR[o] += cR * u; rL[o] += cL * u;
            }
        }
        __threadfence_block(); __syncthreads();
    }
    for (int e = threadIdx.x; e < ps; e += 256) {
        XR[(size_t)n * hstep + (size_t)(e >> 4) * PWc + grp * 16 + (e & 15)] = rR[e];
        XL[(size_t)n * hstep + (size_t)(e >> 4) * PWc + grp * 16 + (e & 15)] = rL[e];
    }
}

// Q[n] = XR[n] - XL[n+1] (both already multiplied by L_{n+1}^-1), n = 0..nt-2; dpsi += E
__global__ void k_forcing_combine(const double *__restrict__ XR, const double *__restrict__ XL, double *__restrict__ Q,
                                  size_t hstep, int nt)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (size_t)(nt - 1) * hstep) Q[i] = XR[i] - XL[i + hstep];
}

__global__ void k_add_inplace(double *__restrict__ dst, const double *__restrict__ src, size_t count)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) dst[i] += src[i];
}

extern "C" {

// forcing terms of eval_forward(...; forcing): E, then Q[n] = L_{n+1}^-1 (rhoR_n - rhoL_{n+1})
int qgdk_forcing_terms(const qgdk_ctx *c)
{
    size_t shm = (size_t)(c->m + 2) * c->Np * 16 * sizeof(double);
    if (c->fs_scratch) shm = 0;           // (panels in this workgroup's HBM slab)
    const size_t hstep = (size_t)c->Np * 2 * c->cp;
#define CALL_FT(N) do { if (shm > 64 * 1024) HIPCHK(hipFuncSetAttribute((const void *)k_forcing_terms<N>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm)); \
        hipLaunchKernelGGL((k_forcing_terms<N>), dim3(c->cp / 8, c->nt), dim3(256), shm, c->stream, c->ops, c->tab, c->ff_F, c->ff_E, \
                           c->ff_XR, c->ff_XL, c->cw, c->Np, c->cp, c->n_ops, c->m, c->fs_scratch); } while (0)
    DISPATCH_NOPS(c->n_ops, CALL_FT)
#undef CALL_FT
    HIPCHK(hipGetLastError());
    if ((size_t)c->Np * 16 * sizeof(double) > 64 * 1024)
        HIPCHK(hipFuncSetAttribute((const void *)k_forced_solve, hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)c->Np * 16 * sizeof(double))));
    hipLaunchKernelGGL(k_forced_solve, dim3(c->cp / 8, c->nt, 2), dim3(256), (size_t)c->Np * 16 * sizeof(double), c->stream,
                       c->LinvT, c->ff_XR, c->ff_XL, c->Np, c->cp, 1, c->nt);
    const size_t cnt = (size_t)(c->nt - 1) * hstep;
    hipLaunchKernelGGL(k_forcing_combine, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, c->stream, c->ff_XR, c->ff_XL, c->ff_Q, hstep, c->nt);
    return (int)hipGetLastError();
}

// derivative history with forcing: w_j = D_j w_0 + E_j -- except at the final time, where the reference
// stores the derivatives WITHOUT the forcing (forward_evolution.jl:229-236 calls compute_derivatives!
// without forcing_matrix there; its own comment asks whether that is intended).  Mirrored.
int qgdk_forcing_add_derivs(const qgdk_ctx *c)
{
    const size_t cnt = (size_t)(c->nt - 1) * c->m * c->Np * 2 * c->cp;
    hipLaunchKernelGGL(k_add_inplace, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, c->stream, c->dpsi, c->ff_E, cnt);
    return (int)hipGetLastError();
}

size_t qgdk_forced_lds(int Np, int m) { return (size_t)(2 * m + 2) * Np * 16 * sizeof(double); }

int qgdk_forced_basis(const qgdk_ctx *c)
{
    size_t shm = qgdk_forced_lds(c->Np, c->m);
    if (c->fs_scratch) shm = 0;
#define CALL_FB(N) do { if (shm > 64 * 1024) HIPCHK(hipFuncSetAttribute((const void *)k_forced_basis<N>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm)); \
        hipLaunchKernelGGL((k_forced_basis<N>), dim3(c->cp / 8, c->nt), dim3(256), shm, c->stream, c->ops, c->tab, c->hist, c->dpsi, \
                           c->fs_BR, c->fs_BL, c->cw, c->Np, c->cp, c->n_ops, c->m, c->fs_scratch); } while (0)
    DISPATCH_NOPS(c->n_ops, CALL_FB)
#undef CALL_FB
    HIPCHK(hipGetLastError());
    const int NB = c->n_ops * 2 * c->m;
    if ((size_t)c->Np * 16 * sizeof(double) > 64 * 1024)
        HIPCHK(hipFuncSetAttribute((const void *)k_forced_solve, hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)c->Np * 16 * sizeof(double))));
    hipLaunchKernelGGL(k_forced_solve, dim3(NB * (c->cp / 8), c->nt, 2), dim3(256), (size_t)c->Np * 16 * sizeof(double), c->stream,
                       c->LinvT, c->fs_BR, c->fs_BL, c->Np, c->cp, NB, c->nt);
    return (int)hipGetLastError();
}

} // extern "C"
