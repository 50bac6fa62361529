// qgd_host_alloc.cpp -- host side of the C ABI (include/qgd.h), handles: creation and validation (mirrors the SchrodingerProb constructor, src/SchrodingerProb.jl:73-154), the time grid,
// its scan layout and windows, the setters and getters.
#include "qgd_host.h"

namespace qgdh {

thread_local std::string g_create_error;


int fail(qgd_handle h, int code, const std::string &msg)
{
    if (h) h->err = msg; else g_create_error = msg;
    if (h && code == QGD_ERR_NUMERIC) h->status_dirty = true;      // (the device's status word is set: see tiny_evaluate)
    return code;
}


void free_pool(std::vector<void *> &pool)
{
    for (void *p : pool) (void)hipFree(p);
    pool.clear();
}


void drop_graph(qgd_handle h)
{
    if (h->graph_exec) { (void)hipGraphExecDestroy(h->graph_exec); h->graph_exec = nullptr; }
    if (h->graph) { (void)hipGraphDestroy(h->graph); h->graph = nullptr; }
    h->graph_calls = 0;
}


static void plan_scan(const qgdk_ctx &k, int S_w, int &B0)
{
    // two scan levels: chain length 2*S/B + 2*B/B2 + B2, near its minimum for B ~ S^(2/3), B2 ~ sqrt(2B)
    B0 = (int)std::lround(std::pow((double)S_w, 2.0 / 3.0));
    if (S_w < 24) B0 = 1;
    if (B0 > 64) B0 = 64;
    if (B0 < 1) B0 = 1;
    if (k.Np > 64 && k.Np <= 640) {     // large-N chains: one workgroup (128 KB of LDS) per CU and 32-column tile (16 beyond Np = 288)
        const int ngt = std::max(k.Np / 32, k.cp / 32);
        B0 = std::min(B0, std::max(8, 256 / std::max(ngt, 1)));
    }
}


// Window layout of the time grid for (part_rank, part_world) -- ranks of a multi-GPU partition -- or, on one GPU, for
// `chunks` windows processed one after the other in the same buffers (bounded memory): sets bpr, blocks_glob, scan_blen,
// scan_blocks(2), scan_g and the window [n_off, n_off + nt) of window `win`.
int plan_windows(qgd_handle h, int chunks, int win)
{
    qgdk_ctx &k = h->k;
    const int S = h->nsteps;
    int W = h->part_world, r = h->part_rank;
    if (chunks > 1) {
        // every chunk gets the scan layout of a stand-alone grid of ceil(S / chunks) steps
        const int Sc = (S + chunks - 1) / chunks;
        int B0; plan_scan(k, Sc, B0);
        k.scan_blen = (Sc + B0 - 1) / B0;
        k.bpr = (Sc + k.scan_blen - 1) / k.scan_blen;
        const int wsteps = k.bpr * k.scan_blen;
        W = (S + wsteps - 1) / wsteps;                 // (>= 1 step in the last window by construction)
        r = win;
        k.blocks_glob = k.bpr * W;
        h->chunks_eff = W;
    } else {
        int B0; plan_scan(k, S, B0);
        k.bpr = (B0 + W - 1) / W;
        for (;;) {   // every rank must own at least one non-empty block
            k.blocks_glob = k.bpr * W;
            k.scan_blen = (S + k.blocks_glob - 1) / k.blocks_glob;
            const int nonempty = (S + k.scan_blen - 1) / k.scan_blen;
            if ((W - 1) * k.bpr < nonempty || k.bpr == 1) break;
            k.bpr--;
        }
        h->chunks_eff = 1;
    }
    k.scan_blocks = k.bpr;     // the scan inside a window runs over its own blocks
    k.blk_lo = r * k.bpr; k.blk_hi = k.blk_lo + k.bpr;
    const int s_lo = k.blk_lo * k.scan_blen;
    const int s_hi = std::min(S, k.blk_hi * k.scan_blen);
    if (s_lo >= S) return fail(h, QGD_ERR_UNSUPPORTED, "too few timesteps for this many ranks (a rank would own no step)");
    k.n_off = s_lo; k.nt = s_hi - s_lo + 1; k.nt_glob = S + 1;
    k.dt = k.tf / S;
    if (chunks > 1) {          // a chunk is a stand-alone grid to the scan kernels; only the time index is global
        k.part_rank = 0; k.part_world = 1;
        k.g_n0 = k.n_off;      // (g_nt is set with the control basis, which covers the whole grid)
    } else {
        k.part_rank = h->part_rank; k.part_world = W;
        k.g_n0 = 0; k.g_nt = 0;
    }
    if (k.scan_blocks > 8) {
        int B2 = (int)std::lround(std::sqrt(2.0 * k.scan_blocks));
        k.scan_g = (k.scan_blocks + B2 - 1) / B2;
        k.scan_blocks2 = (k.scan_blocks + k.scan_g - 1) / k.scan_g;
    } else { k.scan_blocks2 = 1; k.scan_g = k.scan_blocks; }
    return QGD_OK;
}


// Allocate (dry = false) or only add up (dry = true, into *bytes) every device buffer whose size follows the window.
static int alloc_window(qgd_handle h, bool dry, size_t *bytes)
{
    qgdk_ctx &k = h->k;
    size_t total = 0;
    int rc = QGD_OK;
    auto A = [&](double **p, size_t count) -> bool {
        total += count * sizeof(double) + 64;
        if (dry) return true;
        rc = dev_alloc(h, h->grid_bufs, p, count);
        return rc == QGD_OK;
    };
    const size_t Np = k.Np, PW = 2 * Np, PWc = 2 * k.cp, nt = k.nt, m = k.m;
    const size_t panel = Np * PW, pl = Np * Np, hstep = Np * PWc;
    const size_t nb = (size_t)k.scan_blocks, W = (size_t)k.part_world, nb2 = (size_t)k.scan_blocks2;
    k.sigma_planes = k.cp / 8;
    if (Np > 64) k.sigma_planes = std::max(k.sigma_planes, qgdk_dense_sigma_planes_max(k.Np, k.cp, k.m));
    bool ok = A(&k.tab, nt * (m + 1) * (size_t)std::max(k.n_ops, 1) * 2) && A(&k.D, nt * m * panel) && A(&k.L, nt * panel) &&
              A(&k.R, nt * panel) && A(&k.LinvA, nt * 2 * pl) && A(&k.LinvT, nt * 2 * pl) && A(&k.Pr, nt * panel) &&
              A(&k.Pc, nt * 2 * pl) && A(&k.hist, nt * hstep) && A(&k.dpsi, nt * m * hstep) && A(&k.forcing, nt * hstep) &&
              A(&k.yhist, nt * hstep) && A(&k.lam, nt * hstep) &&
              A(&k.sigma, (size_t)k.sigma_planes * nt * (size_t)std::max(k.n_ops, 1) * m * 2) &&
              A(&k.gpart, (nt + 64) * (size_t)std::max(k.cp / 8, 1)) &&
              // blocked scan of the sweeps: chain length 2*blen + B; exchange buffers hold every rank's chunk
              A(&k.PiX, 2 * nb * 2 * pl) && A(&k.phiX, (nb + 1) * hstep) && A(&k.RX, W * 4 * pl) && A(&k.phiRX, W * 2 * hstep) &&
              A(&k.wbnd, (W + 1) * hstep) && A(&k.wbndY, (W + 1) * hstep) && A(&k.bnd, (nb + 1) * hstep) &&
              A(&k.bndY, (nb + 1) * hstep) && A(&k.psi0, hstep) && A(&k.zero_panel, hstep) && A(&k.PiC2, nb2 * 2 * pl) &&
              A(&k.PiR2, nb2 * 2 * pl) && A(&k.phi2, nb2 * hstep) && A(&k.bnd2, (nb2 + 1) * hstep) && A(&k.bndY2, (nb2 + 1) * hstep);
    if (!ok) return rc;
    // fused front (qgd_front.h): phi_0, the adjoint sweep's forcing L^-H f, L_N^-H target; the history of phi takes y's buffer
    if (!dry) { k.front = 0; k.phi0 = k.hforc = k.termU = nullptr; k.phist = k.yhist; }
    if (Np == 64 && h->sparse_available && m <= 4 && h->part_world == 1 &&
        (!A(&k.phi0, hstep) || !A(&k.hforc, nt * hstep) || !A(&k.termU, hstep))) return rc;
    // sub-block history pass (qgd_k_chain.hip): only with compiled-size chains, blocks of at least 6 steps
    k.sub_hist = 0; k.sub_n = 0; if (!dry) k.Hmid = k.Qmid = k.SufP = k.SufPhi = nullptr;
    if ((k.Np == 16 || k.Np == 32 || k.Np == 48 || k.Np == 64) && k.scan_blocks2 > 1 && k.scan_g > 2 &&
        (!A(&k.SufP, (nb2 + 1) * (size_t)(k.scan_g - 2) * 2 * pl) || !A(&k.SufPhi, (nb2 + 1) * (size_t)(k.scan_g - 2) * hstep))) return rc;
    if ((k.Np == 16 || k.Np == 32 || k.Np == 48 || k.Np == 64) && k.scan_blen >= 6) {
        k.sub_hist = 1; k.sub_n = (k.scan_blen + 2) / 3 - 1;      // stored products after 3, 6, ... steps
        if (!A(&k.Hmid, nb * (size_t)k.sub_n * 2 * pl) || !A(&k.Qmid, (nb2 + 1) * (size_t)std::max(k.scan_g, 2) * 2 * pl)) return rc;
    }
    // derivative / gradient kernels.  N <= 64: panels in LDS.  N > 64: the GEMM-style kernels of qgd_k_dense.hip
    // (faster than the LDS-panel kernels at every size measured, scripts/mid_n_timing.py), which keep the m seed
    // panels g_j of a time point in HBM.
    if (!dry) { k.panel_scratch = nullptr; k.dense_gemm = 0; k.Afrag = k.Dfrag = k.OpFrag = nullptr; k.Xouter = k.Tlam = k.Xfrag = nullptr; k.binv = nullptr; }
    if (!dry) { k.gpart_on = 0; k.gpart_n = 0; }
    const bool lds_too_small = qgdk_lds_needed(k.Np, k.m, k.n_ops) > 150 * 1024;
    if (Np > 64) {
        if (!A(&k.panel_scratch, nt * m * hstep) || !A(&k.Afrag, nt * m * 2 * pl) || !A(&k.Dfrag, nt * m * 2 * pl) ||
            !A(&k.OpFrag, (size_t)std::max(k.n_ops, 1) * 2 * pl)) return rc;
        // the m panels X_j of the outer-product form of the gradient scalars, where it pays (qgd_k_dense.hip: dense_sigma_form)
        if (Np >= 128 && k.cp >= 64 && (size_t)(m - 1) * Np < (size_t)(m + 1) * k.cp && m >= 1 &&
            (!A(&k.Xouter, nt * m * panel) || !A(&k.Tlam, nt * 2 * panel) || !A(&k.Xfrag, nt * m * panel))) return rc;
        if (!qgd_path("binv_off") && !A(&k.binv, qgdk_dense_inverse_words(k.Np, (int)nt))) return rc;      // block Gauss-Jordan inverse
        if (!dry) {
            if (qgdk_dense_operator_frag(&k)) return fail(h, QGD_ERR_NO_DEVICE, "operator fragment kernel failed to launch");
            k.dense_gemm = 1;
        }
    } else if (lds_too_small) {
        const size_t slabs = nt * (size_t)(k.cp / 8);
        if (!A(&k.panel_scratch, slabs * (size_t)(2 * m + 1) * Np * 16)) return rc;
    }
    // inverse work slabs when the matrix does not fit in LDS
    const size_t need = (3 * Np + 16 + 2 * pl) * sizeof(double);
    if (need > 150 * 1024 || Np > 64) {      // (the blocked kernel for Np > 64 always works in a slab)
        k.inv_batch = 512;
        // (work slab + the 64 pivot rows of a super-step of k_inverse_blocked2, per workgroup)
        if (!A(&k.inv_scratch, (size_t)k.inv_batch * (2 * pl + 64 * 2 * Np))) return rc;
    } else {
        k.inv_batch = 0; if (!dry) k.inv_scratch = nullptr;
    }
    if (h->chunks_eff > 1 && (!A(&h->chunk_state, ((size_t)h->chunks_eff + 1) * hstep) || !A(&h->carry_y, hstep) || !A(&h->scal_scratch, 8))) return rc;
    if (bytes) *bytes = total;
    return QGD_OK;
}


int alloc_grid(qgd_handle h)
{
    qgdk_ctx &k = h->k;
    drop_graph(h);
    h->grid_ready = false;
    free_pool(h->grid_bufs);
    free_pool(h->forced_bufs); h->forced_key = 0;
    free_pool(h->forcing_bufs); h->forcing_key = 0;
    free_pool(h->stage_bufs); h->stage_hist = h->stage_lam = h->stage_f = nullptr;
    h->dlam = h->dlam_scratch = h->stage_lam_full = nullptr;
    h->chunk_state = h->carry_y = h->scal_scratch = nullptr; h->resident_window = 0;
    // ---- how much of the time grid is resident.  Everything (one window) when it fits the budget; else the grid is
    //      processed in `chunks` windows, one after the other in the same buffers (chunked_forward / chunked_adjoint).
    //      Budget: qgd_set_memory_budget, or 70 % of what is free now (the control basis and the reference-layout
    //      staging buffers come on top).
    int chunks = 1;
    int rc;
    if ((rc = plan_windows(h, 1, 0))) return rc;
    // several kernels carry the time point in gridDim.y (at most 65535): a window never holds more time points than that
    const int MAX_WINDOW_STEPS = 65000;
    if (h->part_world == 1 && !h->comm && !h->comm_pending && h->nsteps > MAX_WINDOW_STEPS) {
        chunks = (h->nsteps + MAX_WINDOW_STEPS - 1) / MAX_WINDOW_STEPS;
        if ((rc = plan_windows(h, chunks, 0))) return rc;
    } else if (k.nt > MAX_WINDOW_STEPS + 500) {
        return fail(h, QGD_ERR_UNSUPPORTED, "a rank's window of the time grid is limited to 65000 steps (use more ranks or one handle with windows)");
    }
    if (h->part_world == 1 && !h->comm && !h->comm_pending) {
        size_t budget = h->mem_budget, fr = 0, tot = 0;
        if (!budget && hipMemGetInfo(&fr, &tot) == hipSuccess) budget = (size_t)(0.7 * (double)fr);
        size_t need = 0;
        (void)alloc_window(h, true, &need);
        const int S = h->nsteps;
        while (budget && need > budget) {
            if (chunks >= S) return fail(h, QGD_ERR_MEMORY, "one time step of this problem does not fit the memory budget (" +
                                                             std::to_string(need) + " bytes needed, " + std::to_string(budget) + " allowed)");
            // the need is close to linear in the window length: jump, then verify
            chunks = std::min<long long>(S, std::max<long long>(chunks + 1, (long long)std::ceil((double)chunks * (double)need / (double)budget)));
            if ((rc = plan_windows(h, chunks, 0))) return rc;
            (void)alloc_window(h, true, &need);
        }
    } else if (h->mem_budget) {
        size_t need = 0;
        (void)alloc_window(h, true, &need);
        if (need > h->mem_budget) return fail(h, QGD_ERR_MEMORY, "a partitioned handle keeps its whole window resident: " + std::to_string(need) +
                                                                 " bytes needed, budget " + std::to_string(h->mem_budget));
    }
    h->chunks_req = chunks;
    if ((rc = alloc_window(h, false, &h->window_bytes))) return rc;
    const size_t hstep = (size_t)k.Np * 2 * k.cp, nt = k.nt;
    HIP_TRY(h, hipMemsetAsync(k.zero_panel, 0, hstep * sizeof(double), k.stream));
    HIP_TRY(h, hipMemsetAsync(k.gpart, 0, (nt + 64) * (size_t)std::max(k.cp / 8, 1) * sizeof(double), k.stream));
    HIP_TRY(h, hipMemcpyAsync(k.bnd2, h->u0v0_panel.data(), hstep * sizeof(double), hipMemcpyHostToDevice, k.stream));
    HIP_TRY(h, hipMemcpyAsync(k.psi0, h->u0v0_panel.data(), hstep * sizeof(double), hipMemcpyHostToDevice, k.stream));
    HIP_TRY(h, hipMemcpyAsync(k.bnd, h->u0v0_panel.data(), hstep * sizeof(double), hipMemcpyHostToDevice, k.stream));
    HIP_TRY(h, hipMemsetAsync(k.hist, 0, nt * hstep * sizeof(double), k.stream));
    HIP_TRY(h, hipMemsetAsync(k.yhist, 0, nt * hstep * sizeof(double), k.stream));
    HIP_TRY(h, hipMemsetAsync(k.lam, 0, nt * hstep * sizeof(double), k.stream));
    HIP_TRY(h, hipMemsetAsync(k.forcing, 0, nt * hstep * sizeof(double), k.stream));
    if (k.hforc) HIP_TRY(h, hipMemsetAsync(k.hforc, 0, nt * hstep * sizeof(double), k.stream));      // (written only under a guard projector)
    h->forcing_zero = true;
    if (k.blk_lo == 0)   // the first window starts at the (constant) initial state
        HIP_TRY(h, hipMemcpyAsync(k.hist, h->u0v0_panel.data(), hstep * sizeof(double), hipMemcpyHostToDevice, k.stream));
    if (h->chunk_state) HIP_TRY(h, hipMemcpyAsync(h->chunk_state, h->u0v0_panel.data(), hstep * sizeof(double), hipMemcpyHostToDevice, k.stream));
    // Hermite weights c_j dt^j and c_j (-dt)^j  (hermite.jl:398-399, :422-423)
    for (int j = 0; j <= k.m; j++) {
        double cj = hermite_coefficient(j, k.m, k.m);
        k.cw_host[2 * j] = cj * std::pow(k.dt, j);
        k.cw_host[2 * j + 1] = cj * std::pow(-k.dt, j);
    }
    HIP_TRY(h, hipMemcpyAsync(k.cw, k.cw_host, sizeof(double) * 2 * (k.m + 1), hipMemcpyHostToDevice, k.stream));
    HIP_TRY(h, hipStreamSynchronize(k.stream));
    h->have_basis = h->have_tables = h->forward_valid = h->derivs_valid = false;
    h->tab_p_host.clear(); h->tab_q_host.clear();
    free_pool(h->basis_bufs);
    k.scal = h->scal_static; k.grad = nullptr; k.redbuf = nullptr; if (h->status_static) k.status = h->status_static;
    h->grid_ready = true;
    return QGD_OK;
}

}  // namespace qgdh

using namespace qgdh;

extern "C" {


int qgd_abi_version(void) { return QGD_ABI_VERSION; }


const char *qgd_last_error(qgd_handle h) { return h ? h->err.c_str() : g_create_error.c_str(); }


int qgd_create(const qgd_problem_desc *d, qgd_handle *out)
{
    if (out) *out = nullptr;
    if (!d || !out) return fail(nullptr, QGD_ERR_ARGUMENT, "null argument");
    const int N = d->N, c = d->n_cols, n_ops = d->n_ops;
    if (N < 1 || c < 1 || n_ops < 0) return fail(nullptr, QGD_ERR_ARGUMENT, "N, n_cols must be positive and n_ops non-negative");
    if (n_ops > QGD_MAX_OPS_DEV) return fail(nullptr, QGD_ERR_UNSUPPORTED, "more than 8 control operators");
    if (d->order < 2 || d->order > QGD_MAX_ORDER || (d->order & 1)) return fail(nullptr, QGD_ERR_ARGUMENT, "order must be even, 2..16");
    if (d->nsteps < 1 || !(d->tf > 0)) return fail(nullptr, QGD_ERR_ARGUMENT, "nsteps and tf must be positive");
    if (!d->system_sym || !d->system_asym || !d->u0 || !d->v0 || (n_ops && (!d->sym_ops || !d->asym_ops)))
        return fail(nullptr, QGD_ERR_ARGUMENT, "null operator or initial-condition pointer");
    if (d->n_ess > N) return fail(nullptr, QGD_ERR_ARGUMENT, "Number of essential levels cannot be greater than the total number of levels.");
    // symmetry checks, SchrodingerProb.jl:73-95
    auto sym_ok = [&](const double *A, double sgn) {
        for (int j = 0; j < N; j++) for (int i = 0; i < N; i++) if (A[i + (size_t)N * j] != sgn * A[j + (size_t)N * i]) return false;
        return true;
    };
    if (!sym_ok(d->system_sym, 1.0)) return fail(nullptr, QGD_ERR_ARGUMENT, "Real part of system Hamiltonian is not symmetric.");
    if (!sym_ok(d->system_asym, -1.0)) return fail(nullptr, QGD_ERR_ARGUMENT, "Imaginary part of system Hamiltonian is not anti-symmetric.");
    for (int o = 0; o < n_ops; o++) {
        if (!sym_ok(d->sym_ops + (size_t)o * N * N, 1.0)) return fail(nullptr, QGD_ERR_ARGUMENT, "Symmetric operator " + std::to_string(o + 1) + " is not symmetric.");
        if (!sym_ok(d->asym_ops + (size_t)o * N * N, -1.0)) return fail(nullptr, QGD_ERR_ARGUMENT, "Anti-symmetric operator " + std::to_string(o + 1) + " is not anti-symmetric.");
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(nullptr, QGD_ERR_NO_DEVICE, "no HIP device available (this library has no CPU fallback)");
    if (d->device < 0 || d->device >= ndev) return fail(nullptr, QGD_ERR_ARGUMENT, "device ordinal out of range");

    qgd_handle h = new qgd_handle_s();
    qgdk_ctx &k = h->k;
    h->device = d->device;
    h->order = d->order;
    k.N = N; k.Np = (N + 15) / 16 * 16; k.c = c; k.cp = (c + 7) / 8 * 8;
    k.n_ops = n_ops; k.n_ess = d->n_ess; k.m = d->order / 2;
    h->nsteps = d->nsteps; k.tf = d->tf; k.dt = d->tf / d->nsteps; k.nt = d->nsteps + 1;
    if ((size_t)2 * k.Np * 16 * sizeof(double) > 150 * 1024) {
        delete h;
        return fail(nullptr, QGD_ERR_UNSUPPORTED, "N too large for the sweep kernels of this version (N <= 592: padded to 16 rows, two 16-column panels in LDS)");
    }
#define CREATE_TRY(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) { std::string m_ = std::string(#expr) + ": " + hipGetErrorString(e__); qgd_destroy(h); return fail(nullptr, QGD_ERR_NO_DEVICE, m_); } } while (0)
#define CREATE_RC(expr) do { int rc__ = (expr); if (rc__) { std::string m_ = h->err; qgd_destroy(h); return fail(nullptr, rc__, m_); } } while (0)
    CREATE_TRY(hipSetDevice(d->device));
    CREATE_TRY(hipStreamCreate(&k.stream));
    const size_t Np = k.Np, pl = Np * Np, PWc = 2 * k.cp;
    // operators: column-major padded planes K_sys, S_sys, Asym_1, Sym_1, ...
    std::vector<double> ops((2 + 2 * (size_t)n_ops) * pl, 0.0);
    auto put = [&](size_t slot, const double *A) {
        for (int j = 0; j < N; j++) for (int i = 0; i < N; i++) ops[slot * pl + i + Np * j] = A[i + (size_t)N * j];
    };
    put(0, d->system_asym); put(1, d->system_sym);
    for (int o = 0; o < n_ops; o++) { put(2 + 2 * o, d->asym_ops + (size_t)o * N * N); put(3 + 2 * o, d->sym_ops + (size_t)o * N * N); }
    CREATE_RC(dev_alloc(h, h->static_bufs, &k.ops, ops.size()));
    CREATE_TRY(hipMemcpy(k.ops, ops.data(), ops.size() * sizeof(double), hipMemcpyHostToDevice));
    // sparse-operator path: ELL over the union pattern of all planes + one ELL list per control.
    // Used when a row of A has at most 16 and at most Np/2 entries (the drift + a_k +/- a_k^dagger
    // operators of multi_qudit_systems.jl have 1 + 2*subsystems); QGD_PATHS=dense_ops keeps the MFMA path.
    {
        const size_t planes = 2 + 2 * (size_t)n_ops;
        std::vector<std::vector<int>> cols(Np);
        int Z = 1, Zo = 1;
        for (size_t r = 0; r < Np; r++) {
            for (size_t cidx = 0; cidx < Np; cidx++) {
                bool nz = false;
                for (size_t q = 0; q < planes && !nz; q++) nz = ops[q * pl + r + Np * cidx] != 0.0;
                if (nz) cols[r].push_back((int)cidx);
            }
            Z = std::max(Z, (int)cols[r].size());
            for (int o = 0; o < n_ops; o++) {
                int cnt = 0;
                for (int cidx : cols[r]) cnt += (ops[(2 + 2 * (size_t)o) * pl + r + Np * cidx] != 0.0 || ops[(3 + 2 * (size_t)o) * pl + r + Np * cidx] != 0.0);
                Zo = std::max(Zo, cnt);
            }
        }
        k.ell_z = Z; k.op_z = Zo;
        h->sparse_available = Np <= 64 && Z <= 16 && 2 * (size_t)Z <= Np && qgdk_sparse_supported(k.Np, k.m, n_ops, Z);
        k.use_sparse = h->sparse_available && !qgd_path("dense_ops");
        if (h->sparse_available) {
            // Slot order.  Default: the e-th nonzero of each row.  When the union pattern is banded -- as many distinct
            // offsets (column - row) as the fullest row has entries, which is the case for the drift + a_k +/- a_k^dagger
            // operators (offsets 0, +-1, +-4, +-16 for subsystems (4,4,4)) -- slot e holds the SAME offset in every row
            // and absent entries point at (row + offset) mod Np with value 0.  The kernels read the neighbour row of
            // slot e for 64 consecutive rows in one ds_read_b128: with one common shift the 16 lanes of a bank group hit
            // 16 different 16-byte slots; with per-row packing the shifts differ from lane to lane and half of the
            // LDS cycles of k_build_LR_ell / k_gradpoint_ell were bank conflicts (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE).
            auto diagonal_slots = [&](const std::vector<std::vector<int>> &cl, int zmax, std::vector<int> &offs) {
                offs.clear();
                for (size_t r = 0; r < Np; r++)
                    for (int cidx : cl[r]) {
                        const int s_ = cidx - (int)r;
                        if (std::find(offs.begin(), offs.end(), s_) == offs.end()) offs.push_back(s_);
                    }
                std::sort(offs.begin(), offs.end());
                return !offs.empty() && (int)offs.size() <= zmax && !qgd_path("ell_row_packed");
            };
            // slots[r][e] = column of slot e in row r, present[r][e] = it is a stored entry
            auto make_slots = [&](const std::vector<std::vector<int>> &cl, int zmax, std::vector<int> &sc, std::vector<char> &pr) {
                std::vector<int> offs;
                sc.assign((size_t)zmax * Np, 0); pr.assign((size_t)zmax * Np, 0);
                if (diagonal_slots(cl, zmax, offs)) {
                    for (size_t r = 0; r < Np; r++)
                        for (int e = 0; e < zmax; e++) {
                            const int s_ = e < (int)offs.size() ? offs[e] : 0;
                            const int cidx = (int)r + s_;
                            const bool have = e < (int)offs.size() && cidx >= 0 && cidx < (int)Np &&
                                              std::find(cl[r].begin(), cl[r].end(), cidx) != cl[r].end();
                            sc[(size_t)e * Np + r] = have ? cidx : (int)(((long long)r + s_ + 4 * (long long)Np) % (long long)Np);
                            pr[(size_t)e * Np + r] = have;
                        }
                } else {
                    for (size_t r = 0; r < Np; r++)
                        for (int e = 0; e < zmax; e++) {
                            const bool have = e < (int)cl[r].size();
                            sc[(size_t)e * Np + r] = have ? cl[r][e] : (int)r;
                            pr[(size_t)e * Np + r] = have;
                        }
                }
            };
            std::vector<int32_t> ecol((size_t)Z * Np), ocol((size_t)std::max(n_ops, 1) * Zo * Np);
            std::vector<double> eval(planes * Z * Np, 0.0), oval((size_t)std::max(n_ops, 1) * 2 * Zo * Np, 0.0);
            std::vector<uint8_t> einv(Np * Np, 0xff);
            std::vector<int> sc; std::vector<char> pr;
            make_slots(cols, Z, sc, pr);
            for (size_t r = 0; r < Np; r++)
                for (int e = 0; e < Z; e++) {
                    const int cidx = sc[(size_t)e * Np + r];
                    ecol[(size_t)e * Np + r] = cidx;
                    if (pr[(size_t)e * Np + r]) {
                        for (size_t q = 0; q < planes; q++) eval[(q * Z + e) * Np + r] = ops[q * pl + r + Np * cidx];
                        einv[r * Np + cidx] = (uint8_t)e;
                    }
                }
            for (int o = 0; o < n_ops; o++) {
                std::vector<std::vector<int>> ocl(Np);
                for (size_t r = 0; r < Np; r++)
                    for (int cidx : cols[r])
                        if (ops[(2 + 2 * (size_t)o) * pl + r + Np * cidx] != 0.0 || ops[(3 + 2 * (size_t)o) * pl + r + Np * cidx] != 0.0)
                            ocl[r].push_back(cidx);
                make_slots(ocl, Zo, sc, pr);
                for (size_t r = 0; r < Np; r++)
                    for (int e = 0; e < Zo; e++) {
                        const int cidx = sc[(size_t)e * Np + r];
                        ocol[((size_t)o * Zo + e) * Np + r] = cidx;
                        if (pr[(size_t)e * Np + r]) {
                            oval[(((size_t)2 * o) * Zo + e) * Np + r] = ops[(2 + 2 * (size_t)o) * pl + r + Np * cidx];
                            oval[(((size_t)2 * o + 1) * Zo + e) * Np + r] = ops[(3 + 2 * (size_t)o) * pl + r + Np * cidx];
                        }
                    }
            }
            CREATE_RC(dev_alloc(h, h->static_bufs, &k.ell_inv, einv.size()));
            CREATE_TRY(hipMemcpy(k.ell_inv, einv.data(), einv.size(), hipMemcpyHostToDevice));
            CREATE_RC(dev_alloc(h, h->static_bufs, &k.ell_col, ecol.size()));
            CREATE_RC(dev_alloc(h, h->static_bufs, &k.ell_val, eval.size()));
            CREATE_RC(dev_alloc(h, h->static_bufs, &k.op_col, ocol.size()));
            CREATE_RC(dev_alloc(h, h->static_bufs, &k.op_val, oval.size()));
            CREATE_TRY(hipMemcpy(k.ell_col, ecol.data(), ecol.size() * sizeof(int32_t), hipMemcpyHostToDevice));
            CREATE_TRY(hipMemcpy(k.ell_val, eval.data(), eval.size() * sizeof(double), hipMemcpyHostToDevice));
            CREATE_TRY(hipMemcpy(k.op_col, ocol.data(), ocol.size() * sizeof(int32_t), hipMemcpyHostToDevice));
            CREATE_TRY(hipMemcpy(k.op_val, oval.data(), oval.size() * sizeof(double), hipMemcpyHostToDevice));
        }
    }
    // guard projector
    k.have_guard = 0;
    if (d->guard) for (size_t e = 0; e < (size_t)4 * N * N; e++) if (d->guard[e] != 0.0) { k.have_guard = 1; break; }
    CREATE_RC(dev_alloc(h, h->static_bufs, &k.guard, (size_t)4 * N * N));
    if (d->guard) CREATE_TRY(hipMemcpy(k.guard, d->guard, sizeof(double) * 4 * N * N, hipMemcpyHostToDevice));
    CREATE_RC(dev_alloc(h, h->static_bufs, &k.guard_diag, (size_t)2 * N));
    if (k.have_guard) {   // diagonal projector (what guard_projector builds): elementwise fast path
        bool diag = true;
        const size_t n2 = 2 * (size_t)N;
        for (size_t j = 0; j < n2 && diag; j++) for (size_t i = 0; i < n2; i++) if (i != j && d->guard[i + n2 * j] != 0.0) { diag = false; break; }
        if (diag) {
            std::vector<double> wd(n2);
            for (size_t i = 0; i < n2; i++) wd[i] = d->guard[i + n2 * i];
            CREATE_TRY(hipMemcpy(k.guard_diag, wd.data(), n2 * sizeof(double), hipMemcpyHostToDevice));
            k.have_guard = 2;
        }
    }
    // initial condition panel
    h->u0v0_panel.assign(Np * PWc, 0.0);
    for (int col = 0; col < c; col++) for (int i = 0; i < N; i++) {
        size_t o = panel_index(i, col, (int)PWc);
        h->u0v0_panel[o] = d->u0[i + (size_t)N * col];
        h->u0v0_panel[o + 8] = d->v0[i + (size_t)N * col];
    }
    CREATE_RC(dev_alloc(h, h->static_bufs, &k.target, Np * PWc));
    CREATE_TRY(hipMemset(k.target, 0, Np * PWc * sizeof(double)));
    CREATE_RC(dev_alloc(h, h->static_bufs, &h->scal_static, (size_t)8));
    k.scal = h->scal_static;
    CREATE_RC(dev_alloc(h, h->static_bufs, &k.term_part, (size_t)2 * 1024 + 2));
    (void)hipMemset(k.term_part, 0, ((size_t)2 * 1024 + 2) * sizeof(double));       // (the ticket counter behind the partial sums)
    CREATE_RC(dev_alloc(h, h->static_bufs, &k.cw, (size_t)2 * 20));
    CREATE_RC(dev_alloc(h, h->static_bufs, &k.status, (size_t)4));      // [singular flag | matrices redone | N = 64: of these, by the last resort | N = 64: first attempt (qgd_inverse_cb.h)]
    h->status_static = k.status;
    // QGD_CREATE_DEFER_GRID: the caller is about to change the grid's layout (qgd_comm_init_rccl / qgd_set_partition /
    // qgd_set_nsteps / qgd_set_memory_budget) -- a rank of a time partition never allocates the WHOLE grid first
    if (!(d->reserved & QGD_CREATE_DEFER_GRID)) CREATE_RC(alloc_grid(h));
    *out = h;
    return QGD_OK;
}


int qgd_create_csc(const qgd_problem_desc *d, const qgd_csc *ssym, const qgd_csc *sasym, const qgd_csc *sym_ops,
                   const qgd_csc *asym_ops, qgd_handle *out)
{
    if (out) *out = nullptr;
    if (!d || !out || !ssym || !sasym || (d->n_ops > 0 && (!sym_ops || !asym_ops))) return fail(nullptr, QGD_ERR_ARGUMENT, "null argument");
    if (d->N < 1 || d->n_ops < 0) return fail(nullptr, QGD_ERR_ARGUMENT, "N, n_cols must be positive and n_ops non-negative");
    const size_t N = (size_t)d->N, nn = N * N, n_ops = (size_t)d->n_ops;
    std::vector<double> dense((2 + 2 * n_ops) * nn, 0.0);
    auto expand = [&](const qgd_csc &a, double *dst) -> bool {
        if (!a.colptr || (a.index_base != 0 && a.index_base != 1)) return false;
        const int64_t b = a.index_base;
        if (a.colptr[0] != b) return false;
        for (size_t j = 0; j < N; j++) {
            if (a.colptr[j + 1] < a.colptr[j]) return false;
            for (int64_t e = a.colptr[j] - b; e < a.colptr[j + 1] - b; e++) {
                if (!a.rowval || !a.nzval) return false;
                const int64_t i = a.rowval[e] - b;
                if (i < 0 || i >= (int64_t)N) return false;
                dst[(size_t)i + N * j] += a.nzval[e];
            }
        }
        return true;
    };
    bool ok = expand(*ssym, dense.data()) && expand(*sasym, dense.data() + nn);
    for (size_t o = 0; o < n_ops && ok; o++)
        ok = expand(sym_ops[o], dense.data() + (2 + o) * nn) && expand(asym_ops[o], dense.data() + (2 + n_ops + o) * nn);
    if (!ok) return fail(nullptr, QGD_ERR_ARGUMENT, "malformed CSC operator (colptr/rowval out of range or index_base not 0/1)");
    qgd_problem_desc dd = *d;
    dd.system_sym = dense.data(); dd.system_asym = dense.data() + nn;
    dd.sym_ops = dense.data() + 2 * nn; dd.asym_ops = dense.data() + (2 + n_ops) * nn;
    return qgd_create(&dd, out);
}


void qgd_destroy(qgd_handle h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream_dead) {
        // a leaked communicator's collective is stuck on the stream (qgd_host_comm.cpp: comm_abort without ncclCommAbort): waiting
        // for the stream, or freeing device memory (which waits for the device), would never return -- the device side of the
        // handle is leaked with the communicator; the host is expected to end the process
        delete h;
        return;
    }
    if (h->k.stream) (void)hipStreamSynchronize(h->k.stream);
    drop_graph(h);
    if (h->comm) (void)qgd_comm_destroy(h);
    if (h->copy_stream) { (void)hipStreamSynchronize(h->copy_stream); (void)hipStreamDestroy(h->copy_stream); }
    if (h->copy_stream2) { (void)hipStreamSynchronize(h->copy_stream2); (void)hipStreamDestroy(h->copy_stream2); }
    if (h->ev_ready) (void)hipEventDestroy(h->ev_ready);
    for (auto &r : h->regs) (void)hipHostUnregister(r.host);
    free_pool(h->stage_bufs);
    free_pool(h->static_bufs); free_pool(h->grid_bufs); free_pool(h->basis_bufs); free_pool(h->forced_bufs); free_pool(h->forcing_bufs);
    for (auto &p : h->phases) { (void)hipEventDestroy(p.e0); (void)hipEventDestroy(p.e1); }
    if (h->host_out) (void)hipHostFree(h->host_out);
    if (h->host_in) (void)hipHostFree(h->host_in);
    if (h->mirror_host) (void)hipHostFree(h->mirror_host);
    if (h->mirror_ticket) (void)hipFree(h->mirror_ticket);
    if (h->k.stream && h->own_stream) (void)hipStreamDestroy(h->k.stream);
    delete h;
}


int qgd_set_nsteps(qgd_handle h, int32_t nsteps, double tf)
{
    if (h) drop_graph(h);
    if (!h) return QGD_ERR_ARGUMENT;
    if (nsteps < 1 || !(tf > 0)) return fail(h, QGD_ERR_ARGUMENT, "nsteps and tf must be positive");
    HIP_TRY(h, hipSetDevice(h->device));
    h->nsteps = nsteps; h->k.tf = tf;
    return alloc_grid(h);
}


int qgd_set_target(qgd_handle h, const double *target_real)
{
    if (h) drop_graph(h);
    if (!h || !target_real) return fail(h, QGD_ERR_ARGUMENT, "null argument");
    HIP_TRY(h, hipSetDevice(h->device));
    qgdk_ctx &k = h->k;
    const size_t PWc = 2 * k.cp;
    std::vector<double> t((size_t)k.Np * PWc, 0.0);
    for (int col = 0; col < k.c; col++) for (int i = 0; i < k.N; i++) {
        size_t o = panel_index(i, col, (int)PWc);
        t[o] = target_real[i + (size_t)2 * k.N * col];
        t[o + 8] = target_real[k.N + i + (size_t)2 * k.N * col];
    }
    HIP_TRY(h, hipMemcpy(k.target, t.data(), t.size() * sizeof(double), hipMemcpyHostToDevice));
    h->target_host.assign(target_real, target_real + (size_t)2 * k.N * k.c);
    k.have_target = 1;
    // (a stored forward sweep of the fused front carries L_N^-H target with it: a history_precomputed call after a new target
    //  redoes the sweep instead of reusing it)
    if (h->front_last) h->fwd_pcof.clear();
    return QGD_OK;
}


int qgd_set_cost_type(qgd_handle h, int32_t cost_type)
{
    if (!h) return fail(h, QGD_ERR_ARGUMENT, "null handle");
    if (cost_type < QGD_COST_INFIDELITY || cost_type > QGD_COST_NORM)
        return fail(h, QGD_ERR_ARGUMENT, "Invalid cost type (0 :Infidelity, 1 :Tracking, 2 :Norm)");    // the reference throws "Invalid cost type"
    if (h->k.cost_type == cost_type) return QGD_OK;      // (a shim that sets it on every call must not cost a captured graph)
    drop_graph(h);
    h->k.cost_type = cost_type;     // (a stored forward sweep stays valid: history_precomputed re-forms the terminal condition)
    if (h->front_last) h->fwd_pcof.clear();      // (... unless it is the fused front's, whose k_psi formed the terminal value of the OLD cost type: the sweep is redone)
    return QGD_OK;
}


int qgd_set_control_basis(qgd_handle h, const int32_t *n_coeff, const double *const *Gp, const double *const *Gq)
{
    if (h) drop_graph(h);
    if (!h || (h->k.n_ops && (!n_coeff || !Gp || !Gq))) return fail(h, QGD_ERR_ARGUMENT, "null argument");
    HIP_TRY(h, hipSetDevice(h->device));
    NEED_GRID(h);
    qgdk_ctx &k = h->k;
    free_pool(h->basis_bufs);
    free_pool(h->forced_bufs); h->forced_key = 0;
    h->have_basis = false; h->forward_valid = false; general_history(h); h->derivs_valid = false; h->fwd_pcof.clear();
    k.scal = h->scal_static; k.grad = nullptr; k.redbuf = nullptr; if (h->status_static) k.status = h->status_static;
    h->ncoef.assign(k.n_ops, 0); h->poff.assign(k.n_ops, 0); h->goff.assign(k.n_ops, 0);
    size_t total = 0; int np = 0, ncmax = 0;
    // (a chunked time grid: the basis covers the WHOLE grid, the kernels of a window index it from the window's offset)
    k.g_nt = 0;
    if (h->chunks_eff > 1) { int rcw = plan_windows(h, h->chunks_req, 0); if (rcw) return rcw; k.g_nt = k.nt_glob; }
    const size_t per = (size_t)(h->chunks_eff > 1 ? k.nt_glob : k.nt) * (k.m + 1);
    for (int o = 0; o < k.n_ops; o++) {
        if (n_coeff[o] < 0) return fail(h, QGD_ERR_ARGUMENT, "negative coefficient count");
        h->ncoef[o] = n_coeff[o]; h->poff[o] = np; h->goff[o] = (int64_t)total;
        if (o < QGD_MAX_OPS_DEV) { k.ncoef_host[o] = n_coeff[o]; k.poff_host[o] = np; k.goff_host[o] = (int64_t)total; }
        np += n_coeff[o]; total += 2 * per * n_coeff[o];
        if (n_coeff[o] > ncmax) ncmax = n_coeff[o];
    }
    k.n_pcof = np; k.nc_max = ncmax;
    int rc;
    if ((rc = dev_alloc(h, h->basis_bufs, &k.G, total + 1))) return rc;
    if ((rc = dev_alloc(h, h->basis_bufs, &k.goff, (size_t)k.n_ops + 1))) return rc;
    if ((rc = dev_alloc(h, h->basis_bufs, &k.ncoef, (size_t)k.n_ops + 1))) return rc;
    if ((rc = dev_alloc(h, h->basis_bufs, &k.poff, (size_t)k.n_ops + 1))) return rc;
    if ((rc = dev_alloc(h, h->basis_bufs, &h->pcof_dev, (size_t)np + 1))) return rc;
    if ((rc = dev_alloc(h, h->basis_bufs, &k.redbuf, (size_t)np + 8))) return rc;
    if ((rc = dev_alloc(h, h->basis_bufs, &h->redglob, (size_t)np + 8))) return rc;
    {   // k_contract: per-time-chunk partial sums, added in chunk order by k_contract_sum
        // (rows: the time chunks of k_contract, or -- sparse path -- one per (column group, time point) from k_gradpoint_ell)
        const size_t chunks = std::max(((size_t)k.nt + 7) / 8, (size_t)(k.cp / 8) * (size_t)k.nt);
        if ((rc = dev_alloc(h, h->basis_bufs, &k.cpart, chunks * (size_t)std::max(np, 1)))) return rc;
    }
    k.grad = k.redbuf; k.scal = k.redbuf + np;     // [grad | scal]: one all-reduce in the multi-GPU path
    k.status = reinterpret_cast<int *>(k.redbuf + np + 4);   // ... and [grad | scal | status]: one copy to the host
    HIP_TRY(h, hipMemset(k.redbuf, 0, ((size_t)np + 8) * sizeof(double)));
    if (h->host_out_len < (size_t)np + 8) {
        if (h->host_out) (void)hipHostFree(h->host_out);
        if (h->host_in) (void)hipHostFree(h->host_in);
        h->host_out = nullptr; h->host_in = nullptr; h->host_out_len = 0;
        HIP_TRY(h, hipHostMalloc((void **)&h->host_out, ((size_t)np + 8) * sizeof(double), hipHostMallocDefault));
        HIP_TRY(h, hipHostMalloc((void **)&h->host_in, ((size_t)np + 8) * sizeof(double), hipHostMallocDefault));
        h->host_out_len = (size_t)np + 8;
        if (h->mirror_host) { (void)hipHostFree(h->mirror_host); h->mirror_host = h->mirror_dev = nullptr; }
        if (!h->mirror_off) {      // (optional: without it the results come back by a copy packet)
            void *dev = nullptr;
            if (hipHostMalloc((void **)&h->mirror_host, ((size_t)np + 8) * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess &&
                hipHostGetDevicePointer(&dev, h->mirror_host, 0) == hipSuccess) {
                h->mirror_dev = static_cast<double *>(dev);
                memset(h->mirror_host, 0, ((size_t)np + 8) * sizeof(double));
            } else {
                (void)hipGetLastError();
                if (h->mirror_host) (void)hipHostFree(h->mirror_host);
                h->mirror_host = h->mirror_dev = nullptr;
            }
        }
    }
    if (!h->mirror_ticket && !h->mirror_off) {
        if (hipMalloc((void **)&h->mirror_ticket, 64) == hipSuccess) (void)hipMemset(h->mirror_ticket, 0, 64);
        else { (void)hipGetLastError(); h->mirror_ticket = nullptr; }
    }
    h->mirror_seq = 0;
    if (h->mirror_host) memset(h->mirror_host, 0, ((size_t)np + 8) * sizeof(double));
    for (int o = 0; o < k.n_ops; o++) {
        const size_t cnt = per * h->ncoef[o];
        if (!cnt) continue;
        HIP_TRY(h, hipMemcpy(k.G + h->goff[o], Gp[o], cnt * sizeof(double), hipMemcpyHostToDevice));
        HIP_TRY(h, hipMemcpy(k.G + h->goff[o] + cnt, Gq[o], cnt * sizeof(double), hipMemcpyHostToDevice));
    }
    if (k.n_ops) {
        HIP_TRY(h, hipMemcpy(k.goff, h->goff.data(), sizeof(int64_t) * k.n_ops, hipMemcpyHostToDevice));
        HIP_TRY(h, hipMemcpy(k.ncoef, h->ncoef.data(), sizeof(int32_t) * k.n_ops, hipMemcpyHostToDevice));
        HIP_TRY(h, hipMemcpy(k.poff, h->poff.data(), sizeof(int32_t) * k.n_ops, hipMemcpyHostToDevice));
    }
    h->have_basis = true;
    return QGD_OK;
}


int qgd_set_control_tables(qgd_handle h, const double *pt, const double *qt)
{
    if (h) drop_graph(h);
    if (!h || !pt || !qt) return fail(h, QGD_ERR_ARGUMENT, "null argument");
    HIP_TRY(h, hipSetDevice(h->device));
    NEED_GRID(h);
    qgdk_ctx &k = h->k;
    if (h->chunks_eff > 1) {      // a windowed grid: the tables of the whole grid stay on the host, each window uploads its slice
        const size_t all = (size_t)k.nt_glob * (k.m + 1) * k.n_ops;
        h->tab_p_host.assign(pt, pt + all); h->tab_q_host.assign(qt, qt + all);
        h->have_tables = true; h->forward_valid = false; general_history(h);
        return QGD_OK;
    }
    h->tab_p_host.clear(); h->tab_q_host.clear();
    const size_t cnt = (size_t)k.nt * (k.m + 1) * k.n_ops;
    double *tmp = nullptr;
    HIP_TRY(h, hipMalloc((void **)&tmp, 2 * cnt * sizeof(double) + 64));
    hipError_t e1 = hipMemcpy(tmp, pt, cnt * sizeof(double), hipMemcpyHostToDevice);
    hipError_t e2 = hipMemcpy(tmp + cnt, qt, cnt * sizeof(double), hipMemcpyHostToDevice);
    int kr = (e1 == hipSuccess && e2 == hipSuccess) ? qgdk_tables_from_host(&k, tmp, tmp + cnt) : 1;
    (void)hipStreamSynchronize(k.stream);
    (void)hipFree(tmp);
    if (kr) return fail(h, QGD_ERR_NO_DEVICE, "uploading control tables failed");
    h->have_tables = true;
    return QGD_OK;
}


// ---------------------------------------------------------------------------
// Time-partitioned (multi-GPU) evaluation.  The library does no communication itself: the
// caller moves the two exchange buffers and the reduction buffer with its own collectives
// (torch.distributed / RCCL in bench.py, MPI from Julia) between the phases.
// ---------------------------------------------------------------------------
int qgd_set_partition(qgd_handle h, int32_t rank, int32_t world)
{
    if (h) drop_graph(h);
    if (!h) return QGD_ERR_ARGUMENT;
    if (world < 1 || rank < 0 || rank >= world) return fail(h, QGD_ERR_ARGUMENT, "rank/world out of range");
    HIP_TRY(h, hipSetDevice(h->device));
    h->part_rank = rank; h->part_world = world;
    return alloc_grid(h);
}


int qgd_get_partition(qgd_handle h, int32_t *out8)
{
    if (!h || !out8) return QGD_ERR_ARGUMENT;
    NEED_GRID(h);
    const qgdk_ctx &k = h->k;
    out8[0] = k.n_off; out8[1] = k.n_off + k.nt - 1;   // first / last global time point of the window
    // One GPU working through the grid in windows (qgd_set_memory_budget): the windows are the library's business -- the
    // caller owns ALL time points (control basis and reference-layout outputs cover the whole grid); k.n_off / k.nt are
    // whichever window was processed last.
    if (h->chunks_eff > 1 && h->part_world == 1) { out8[0] = 0; out8[1] = k.nt_glob - 1; }
    out8[2] = k.blocks_glob; out8[3] = k.bpr; out8[4] = k.scan_blen; out8[5] = k.part_rank; out8[6] = k.part_world;
    out8[7] = k.nt_glob;
    return QGD_OK;
}


int qgd_set_stream(qgd_handle h, void *stream)
{
    if (h) drop_graph(h);
    if (!h) return QGD_ERR_ARGUMENT;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->k.stream));
    if (h->own_stream && h->k.stream) (void)hipStreamDestroy(h->k.stream);
    h->k.stream = (hipStream_t)stream;
    h->own_stream = false;
    return QGD_OK;
}


int qgd_register_host_buffer(qgd_handle h, void *ptr, size_t bytes)
{
    if (!h || !ptr || !bytes) return fail(h, QGD_ERR_ARGUMENT, "null argument");
    HIP_TRY(h, hipSetDevice(h->device));
    if (find_reg(h, ptr, bytes)) return QGD_OK;
    hipError_t e = hipHostRegister(ptr, bytes, hipHostRegisterMapped);
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(h, QGD_ERR_NO_DEVICE, std::string("hipHostRegister: ") + hipGetErrorString(e)); }
    void *dev = nullptr;
    if (hipHostGetDevicePointer(&dev, ptr, 0) != hipSuccess) { (void)hipGetLastError(); dev = nullptr; }   // (no mapping: staged copies)
    h->regs.push_back({ptr, dev, bytes, false});
    return QGD_OK;
}


int qgd_unregister_host_buffer(qgd_handle h, void *ptr)
{
    if (!h || !ptr) return fail(h, QGD_ERR_ARGUMENT, "null argument");
    HIP_TRY(h, hipSetDevice(h->device));
    for (size_t i = 0; i < h->regs.size(); i++)
        if (h->regs[i].host == ptr) {
            if (h->copy_stream) (void)hipStreamSynchronize(h->copy_stream);
            (void)hipHostUnregister(ptr);
            h->regs.erase(h->regs.begin() + i);
            return QGD_OK;
        }
    return fail(h, QGD_ERR_ARGUMENT, "buffer was not registered");
}


int qgd_set_operator_path(qgd_handle h, int32_t mode)
{
    if (h) drop_graph(h);
    if (!h) return QGD_ERR_ARGUMENT;
    if (mode < 0 || mode > 2) return fail(h, QGD_ERR_ARGUMENT, "operator path: 0 automatic, 1 dense, 2 sparse");
    if (mode == 2 && !h->sparse_available)
        return fail(h, QGD_ERR_UNSUPPORTED, "the operators are too dense (or N > 64) for the sparse kernels");
    h->k.use_sparse = (mode == 2) || (mode == 0 && h->sparse_available && !qgd_path("dense_ops"));
    h->forward_valid = false; general_history(h); h->derivs_valid = false;
    return QGD_OK;
}


int qgd_get_operator_path(qgd_handle h, int32_t *out3)
{
    if (!h || !out3) return QGD_ERR_ARGUMENT;
    out3[0] = h->k.use_sparse ? 2 : 1; out3[1] = h->k.ell_z; out3[2] = h->k.op_z;
    return QGD_OK;
}


int qgd_set_small_path(qgd_handle h, int32_t on)
{
    if (!h) return QGD_ERR_ARGUMENT;
    h->small_path = (on != 0);
    return QGD_OK;
}


int qgd_set_lambda_derivatives(qgd_handle h, int32_t on)
{
    if (!h) return QGD_ERR_ARGUMENT;
    if (h->lambda_derivs == (on != 0)) return QGD_OK;      // unchanged: a registered lambda_history keeps its zero fill
    h->lambda_derivs = (on != 0);
    for (auto &r : h->regs) r.zeroed = false;      // columns 1..m of a registered lambda_history change meaning
    return QGD_OK;
}


int qgd_set_memory_budget(qgd_handle h, size_t bytes)
{
    if (h) drop_graph(h);
    if (!h) return QGD_ERR_ARGUMENT;
    if (h->comm || h->part_world != 1) return fail(h, QGD_ERR_STATE, "a partitioned handle keeps its window resident: set the budget before the partition");
    HIP_TRY(h, hipSetDevice(h->device));
    h->mem_budget = bytes;
    return alloc_grid(h);          // (invalidates control basis and histories, like qgd_set_nsteps)
}


int qgd_get_memory_plan(qgd_handle h, int64_t *out4)
{
    if (!h || !out4) return QGD_ERR_ARGUMENT;
    NEED_GRID(h);
    const qgdk_ctx &k = h->k;
    out4[0] = h->chunks_eff; out4[1] = (int64_t)k.bpr * k.scan_blen; out4[2] = (int64_t)h->window_bytes; out4[3] = (int64_t)h->mem_budget;
    return QGD_OK;
}


int qgd_set_save_every(qgd_handle h, int32_t save_every_nsteps)
{
    if (!h) return QGD_ERR_ARGUMENT;
    if (save_every_nsteps < 1) return fail(h, QGD_ERR_ARGUMENT, "saveEveryNsteps must be a positive integer");
    h->save_every = save_every_nsteps;
    return QGD_OK;
}


int qgd_set_timing(qgd_handle h, int32_t mode, const char *phase)
{
    if (!h) return QGD_ERR_ARGUMENT;
    h->timing = (mode != 0);
    h->timing_only = (mode == 2 && phase) ? phase : "";
    // (turning the bracketing OFF keeps the last recorded pairs readable: a caller can sample one evaluation, switch off and
    //  read the times later, outside its own timed region -- bench.py)
    if (mode != 0) for (auto &p : h->phases) p.used = false;
    return QGD_OK;
}


int qgd_get_timings(qgd_handle h, const char **names, float *ms, int32_t cap, int32_t *n)
{
    if (!h || !n) return QGD_ERR_ARGUMENT;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->k.stream);
    int cnt = 0;
    for (size_t i = 0; i < h->phases.size(); i++) {
        auto &p = h->phases[i];
        if (!p.used) continue;
        bool first = true;        // the pieces of one phase (slots of the time-chunk pipeline) are reported as their sum
        for (size_t j = 0; j < i; j++) if (h->phases[j].used && !strcmp(h->phases[j].name, p.name)) first = false;
        if (!first) continue;
        if (cnt < cap && names && ms) {
            names[cnt] = p.name;
            float tot = 0.f;
            for (size_t j = i; j < h->phases.size(); j++) {
                auto &q = h->phases[j];
                if (!q.used || strcmp(q.name, p.name)) continue;
                float t = 0.f;
                if (hipEventElapsedTime(&t, q.e0, q.e1) != hipSuccess) { tot = -1.f; break; }
                tot += t;
            }
            ms[cnt] = tot;
        }
        cnt++;
    }
    *n = cnt;
    return QGD_OK;
}

}  // extern "C"
