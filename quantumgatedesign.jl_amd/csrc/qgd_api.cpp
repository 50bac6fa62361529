// qgd_api.cpp -- host side of the C ABI declared in include/qgd.h.
// Validation mirrors the SchrodingerProb constructor (src/SchrodingerProb.jl:73-154);
// orchestration mirrors eval_forward! (src/forward_evolution.jl:33-70) and
// discrete_adjoint! (src/eval_grad_discrete_adjoint.jl:107-160).
// There is no CPU fallback: without a GPU every compute entry point fails.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>      // types only: the library is bound at run time (dlopen), see RcclApi
#include <dlfcn.h>
#include <sys/mman.h>
#include <sys/syscall.h>
#include <unistd.h>
#include <cctype>
#include <sched.h>

#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <string>
#include <vector>
#include <algorithm>

#include "qgd.h"
#include "qgd_device.h"

namespace {

thread_local std::string g_create_error;

struct Phase { const char *name; int slot; hipEvent_t e0, e1; bool used; };

}  // namespace

struct qgd_handle_s {
    qgdk_ctx k{};
    int order = 0, nsteps = 0, device = 0;      // nsteps: GLOBAL number of timesteps
    int part_rank = 0, part_world = 1;
    bool own_stream = true;
    bool timing = (getenv("QGD_PHASE_TIMING") != nullptr);   // per-phase HIP events are opt-in (qgd_set_timing): 26 event records cost ~0.17 ms
    std::string timing_only;            // when non-empty: only this phase is bracketed by events
    std::string err;
    std::vector<void *> static_bufs, grid_bufs, basis_bufs, forced_bufs;
    std::vector<double> target_host;   // stacked real target [2N x c] (forced gradient: the overlaps are host arithmetic)
    size_t forced_key = 0;             // (nt, n_pcof) the forced-gradient buffers were sized for
    double *fsc_forced = nullptr, *fsc_forcing = nullptr;   // HBM work-panel slabs of the forced kernels when they exceed the LDS (N > 64)
    std::vector<void *> forcing_bufs;  // eval_forward with a user forcing
    size_t forcing_key = 0;
    bool have_basis = false, have_tables = false, forward_valid = false, derivs_valid = false;
    std::vector<int32_t> ncoef, poff;
    std::vector<int64_t> goff;
    double *pcof_dev = nullptr;
    double *scal_static = nullptr;
    std::vector<Phase> phases;
    std::vector<double> u0v0_panel;   // host copy of the initial panel
    bool sparse_available = false;    // the ELL lists were built and fit the sparse kernels
    int *status_static = nullptr;     // singularity flag when no control basis is set (else it lives in redbuf)
    double *host_out = nullptr;       // pinned staging buffer for [grad | scal | status]: one copy per evaluation
    double *host_in = nullptr;        // pinned staging buffer for pcof (a pageable source makes the upload synchronous)
    size_t host_out_len = 0;
    // result mirror (qgd_device.h): [grad | scal(4) | status | sequence number] in coherent pinned host memory that the last
    // kernel of a gradient evaluation writes itself; the host polls the sequence number instead of waiting for a copy packet
    // and the stream's completion signal.  QGD_RESULT_MIRROR=0 keeps the copy + hipStreamSynchronize.
    double *mirror_host = nullptr, *mirror_dev = nullptr;
    unsigned int *mirror_ticket = nullptr;
    bool status_dirty = false;          // an evaluation ended with the singular-matrix flag set on the device (reset before the small-problem path runs)
    unsigned long long mirror_seq = 0;
    bool mirror_armed = false;          // the evaluation in flight ends with a mirrored k_contract_sum
    // Small problems (N <= 4, <= 4 columns, <= 128 time points: Rabi, the two-qubit CNOT) take the four-launch path of
    // qgd_k_tiny.hip for calls that return only [grad | scalars].  That path leaves none of the general path's intermediates
    // behind: history_stale makes a later call that needs them (history_precomputed with output arrays, qgd_get_intermediate)
    // redo the evaluation on the general path first.  QGD_TINY=0 / qgd_set_small_path(h, 0): off.
    bool small_path = !(getenv("QGD_TINY") && atoi(getenv("QGD_TINY")) == 0);
    bool history_stale = false;
    std::vector<double> tiny_pcof;      // pcof of the last small-path evaluation
    bool tiny_was_gradient = false;
    bool mirror_off = (getenv("QGD_RESULT_MIRROR") && atoi(getenv("QGD_RESULT_MIRROR")) == 0);
    // the launch sequence of one full gradient evaluation as a hipGraph, opt-in (QGD_GRAPH=1).  Measured: no gain
    // on cnot3 (420 us either way) and 5 % on cnot2 (98 vs 104 us) -- an evaluation is a chain of ~15 DEPENDENT
    // kernels and the ~6 us per dependent dispatch is spent on the device side, not in hipLaunchKernel; the
    // instantiation costs several ms once.  Captured on the third eligible call, dropped by every entry point
    // that changes buffers, sizes or options.
    // reference-layout outputs (uv_history, lambda_history, adjoint_forcing): re-laid out on the device
    // (qgd_k_layout.hip) into staging buffers and copied out on a second stream, so that the download of the
    // state history overlaps the adjoint sweep.  Host buffers the caller registered (qgd_register_host_buffer)
    // are pinned: the copies then run at PCIe speed.
    struct HostReg { void *host; void *dev; size_t bytes; bool zeroed; };
    std::vector<HostReg> regs;
    std::vector<void *> stage_bufs;
    double *stage_hist = nullptr, *stage_lam = nullptr, *stage_f = nullptr;
    // qgd_set_lambda_derivatives: the m derivative columns of lambda_history as the reference leaves them
    bool lambda_derivs = false;
    double *dlam = nullptr, *dlam_scratch = nullptr, *stage_lam_full = nullptr;
    hipStream_t copy_stream = nullptr;
    hipStream_t copy_stream2 = nullptr;   // the second half of a large pinned download goes to a second DMA engine (QGD_COPY_SPLIT=0: off)
    hipEvent_t ev_ready = nullptr;
    std::vector<double> fwd_pcof;       // the pcof of the forward sweep that is on the device (history_precomputed)
    std::vector<double> scatter_tmp;    // unregistered lambda_history: compact copy, scattered on the host
    bool copies_pending = false;
    bool forcing_zero = false;          // no guard projector: the adjoint forcing is all zeros and nothing has written it since
                                        // (alloc_grid clears it, qgd_eval_adjoint uploads a caller's forcing into it)
    bool defer_terminal = false;        // a full gradient evaluation: the overlaps and y_N ride in the first adjoint launch
    double *lambda_out = nullptr;       // lambda_history of the evaluation in flight (copied out right after the lambda phase)
    // Time-chunk pipeline of the front of an evaluation (sparse path, N = 64).  build_LR, inverse+propagator and the
    // block products are each latency-bound launches that leave most CUs idle at their tails, and time point n of one
    // needs only time points n-1, n of the one before: the grid is cut into pipe_chunks groups of scan blocks; the
    // build kernels run back to back on the main stream, and chunk i's inverse + block products run on side stream i
    // as soon as its build is done -- beside the build of chunk i+1 and the inverses of the other chunks.
    // Off by default (1 chunk): measured slower on cnot3 ("time chunks over streams");
    // QGD_PIPE_CHUNKS=2..8 turns it on.
    enum { MAX_CHUNKS = 8 };
    int pipe_chunks = getenv("QGD_PIPE_CHUNKS") ? atoi(getenv("QGD_PIPE_CHUNKS")) : 1;
    hipStream_t pipe_stream[MAX_CHUNKS] = {};
    hipEvent_t pipe_built[MAX_CHUNKS] = {}, pipe_done[MAX_CHUNKS] = {};
    hipGraph_t graph = nullptr;
    hipGraphExec_t graph_exec = nullptr;
    int graph_calls = 0;
    bool graph_off = (getenv("QGD_GRAPH") == nullptr);
    // multi-GPU INSIDE the library (qgd_comm_init_rccl): the ranks that share one evaluation talk over an RCCL
    // communicator; qgd_discrete_adjoint / qgd_eval_forward then run the partitioned protocol themselves, the
    // collectives issued on the handle's stream between the phases (no host synchronisation in between).
    ncclComm_t comm = nullptr;
    int comm_shard = QGD_SHARD_TIME, comm_rank = 0, comm_world = 1;
    // failure mode of the collective calls (qgd_set_comm_timeout): the one host wait of a collective evaluation is bounded;
    // when it expires, when RCCL reports an asynchronous error, or when this rank fails locally between two collectives,
    // the communicator is ABORTED (ncclCommAbort: its kernels leave the stream) and the call returns QGD_ERR_COMM --
    // the other ranks then run into their own bound instead of waiting for this one forever.
    double comm_timeout_ms = getenv("QGD_COMM_TIMEOUT_MS") ? atof(getenv("QGD_COMM_TIMEOUT_MS")) : 30000.0;
    int comm_fail_at = 0;               // qgd_comm_debug_fail_at (test hook): pretend a local failure in front of collective #n
    bool grid_ready = false;            // QGD_CREATE_DEFER_GRID: the time grid is allocated by the first entry point that needs it
    bool comm_pending = false;          // qgd_comm_init_rccl is re-allocating the grid for the communicator it is about to
                                        // create: the grid stays resident (comm_discrete_adjoint does not walk windows)
    // bounded-memory time grid (qgd_set_memory_budget): chunks_eff windows of the grid share the per-time-point buffers
    size_t mem_budget = 0;              // bytes; 0 = 70 % of the free device memory when the grid is allocated
    int chunks_eff = 1;                 // windows the grid is processed in (1: everything resident)
    int chunks_req = 1;                 // the count plan_windows derived that layout from
    size_t window_bytes = 0;            // device bytes of the per-window buffers
    int resident_window = 0;            // whose step matrices are in the buffers right now
    double *chunk_state = nullptr;      // [chunks_eff + 1][Np][2cp]: the state at the start of every window (+ the final state)
    double *carry_y = nullptr;          // y at the end of the window the adjoint pass does next
    std::vector<double> tab_p_host, tab_q_host;      // qgd_set_control_tables on a windowed grid: the caller's tables for the WHOLE grid
    double *scal_scratch = nullptr;     // where a re-run of a window's forward sweep puts its guard sum (already counted)
    int save_every = 1;                 // qgd_set_save_every: uv_history of qgd_eval_forward holds every save_every-th time point
    double *redglob = nullptr;          // [n_pcof + 8] time shards: the reductions are out of place (send = the rank's own
                                        // [grad | scalars], receive = this), so a later history_precomputed call still finds
                                        // the rank's OWN guard sum and overlaps on the device, not the sums over the ranks
};

namespace {

int fail(qgd_handle h, int code, const std::string &msg)
{
    if (h) h->err = msg; else g_create_error = msg;
    if (h && code == QGD_ERR_NUMERIC) h->status_dirty = true;      // (the device's status word is set: see tiny_evaluate)
    return code;
}

#define HIP_TRY(h, expr)                                                                    \
    do {                                                                                    \
        hipError_t e__ = (expr);                                                            \
        if (e__ != hipSuccess)                                                              \
            return fail((h), QGD_ERR_NO_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e__)); \
    } while (0)

template <typename T>
int dev_alloc(qgd_handle h, std::vector<void *> &pool, T **p, size_t count)
{
    void *q = nullptr;
    hipError_t e = hipMalloc(&q, count * sizeof(T) + 64);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        // an allocation that does not fit is the caller's problem size, not a missing device (qgd.h: QGD_ERR_MEMORY)
        return fail(h, e == hipErrorOutOfMemory ? QGD_ERR_MEMORY : QGD_ERR_NO_DEVICE,
                    std::string("hipMalloc of ") + std::to_string(count * sizeof(T)) + " bytes: " + hipGetErrorString(e));
    }
    pool.push_back(q);
    *p = static_cast<T *>(q);
    return QGD_OK;
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

double factorial(int n) { double f = 1; for (int i = 2; i <= n; i++) f *= i; return f; }
// hermite.jl:389-391
double hermite_coefficient(int j, int p, int q) { return factorial(p) * factorial(p + q - j) / (factorial(p + q) * factorial(p - j)); }

inline size_t panel_index(int row, int col, int PWc) { return (size_t)row * PWc + (col >> 3) * 16 + (col & 7); }

// One event pair per (phase, slot): the pieces of a phase that the time-chunk pipeline launches on its side streams
// are bracketed separately (slot = chunk) and qgd_get_timings adds them up.
struct PhaseTimer {
    qgd_handle h; size_t idx; bool on; hipStream_t stream;
    PhaseTimer(qgd_handle h_, const char *name, hipStream_t s = nullptr, int slot = 0) : h(h_), idx(0), on(h_->timing), stream(s ? s : h_->k.stream)
    {
        if (on && !h->timing_only.empty() && h->timing_only != name) on = false;
        if (!on) return;
        for (idx = 0; idx < h->phases.size(); idx++) if (!strcmp(h->phases[idx].name, name) && h->phases[idx].slot == slot) break;
        if (idx == h->phases.size()) {
            Phase p{name, slot, nullptr, nullptr, false};
            (void)hipEventCreate(&p.e0); (void)hipEventCreate(&p.e1);
            h->phases.push_back(p);
        }
        h->phases[idx].used = true;
        (void)hipEventRecord(h->phases[idx].e0, stream);
    }
    ~PhaseTimer() { if (on) (void)hipEventRecord(h->phases[idx].e1, stream); }
};

// The scan layout of one window of `S_w` steps: B blocks of blen steps (+ the second level for B > 8).
static void plan_scan(const qgdk_ctx &k, int S_w, int &B0)
{
    // two scan levels: chain length 2*S/B + 2*B/B2 + B2, near its minimum for B ~ S^(2/3), B2 ~ sqrt(2B)
    B0 = (int)std::lround(std::pow((double)S_w, 2.0 / 3.0));
    if (S_w < 24) B0 = 1;
    if (B0 > 64) B0 = 64;
    if (getenv("QGD_SCAN_B0")) B0 = atoi(getenv("QGD_SCAN_B0"));      // (tuning experiments)
    if (B0 < 1) B0 = 1;
    if (k.Np > 64 && k.Np <= 640) {     // large-N chains: one workgroup (128 KB of LDS) per CU and 32-column tile (16 beyond Np = 288)
        const int ngt = std::max(k.Np / 32, k.cp / 32);
        B0 = std::min(B0, std::max(8, 256 / std::max(ngt, 1)));
    }
}

// Window layout of the time grid for (part_rank, part_world) -- ranks of a multi-GPU partition -- or, on one GPU, for
// `chunks` windows processed one after the other in the same buffers (bounded memory): sets bpr, blocks_glob, scan_blen,
// scan_blocks(2), scan_g and the window [n_off, n_off + nt) of window `win`.
static int plan_windows(qgd_handle h, int chunks, int win)
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
        if (getenv("QGD_SCAN_B2")) B2 = std::max(1, atoi(getenv("QGD_SCAN_B2")));
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
    if (Np > 64 && !getenv("QGD_DENSE_OLD")) k.sigma_planes = std::max(k.sigma_planes, qgdk_dense_sigma_planes_max(k.Np, k.cp, k.m));
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
    // sub-block history pass (qgd_k_chain.hip): only with compiled-size chains, blocks of at least 6 steps
    k.sub_hist = 0; k.sub_n = 0; if (!dry) k.Hmid = k.Qmid = k.SufP = k.SufPhi = nullptr;
    if ((k.Np == 16 || k.Np == 32 || k.Np == 48 || k.Np == 64) && k.scan_blocks2 > 1 && k.scan_g > 2 &&
        (!A(&k.SufP, (nb2 + 1) * (size_t)(k.scan_g - 2) * 2 * pl) || !A(&k.SufPhi, (nb2 + 1) * (size_t)(k.scan_g - 2) * hstep))) return rc;
    if ((k.Np == 16 || k.Np == 32 || k.Np == 48 || k.Np == 64) && k.scan_blen >= 6 && !getenv("QGD_HIST_WHOLE_BLOCKS")) {
        k.sub_hist = 1; k.sub_n = (k.scan_blen + 2) / 3 - 1;      // stored products after 3, 6, ... steps
        if (!A(&k.Hmid, nb * (size_t)k.sub_n * 2 * pl) || !A(&k.Qmid, (nb2 + 1) * (size_t)std::max(k.scan_g, 2) * 2 * pl)) return rc;
    }
    // derivative / gradient kernels.  N <= 64: panels in LDS.  N > 64: the GEMM-style kernels of qgd_k_dense.hip
    // (faster than the LDS-panel kernels at every size measured, scripts/mid_n_timing.py), which keep the m seed
    // panels g_j of a time point in HBM.  QGD_DENSE_OLD=1 keeps the older kernels (LDS panels, or HBM slabs when
    // they do not fit) for comparison.
    if (!dry) { k.panel_scratch = nullptr; k.dense_gemm = 0; k.Afrag = k.Dfrag = k.OpFrag = nullptr; k.Xouter = k.Tlam = k.Xfrag = nullptr; k.binv = nullptr; }
    if (!dry) { k.gpart_on = 0; k.gpart_n = 0; }
    const bool lds_too_small = qgdk_lds_needed(k.Np, k.m, k.n_ops) > 150 * 1024 || getenv("QGD_FORCE_GLOBAL_PANELS");
    if (Np > 64 && !getenv("QGD_DENSE_OLD")) {
        if (!A(&k.panel_scratch, nt * m * hstep) || !A(&k.Afrag, nt * m * 2 * pl) || !A(&k.Dfrag, nt * m * 2 * pl) ||
            !A(&k.OpFrag, (size_t)std::max(k.n_ops, 1) * 2 * pl)) return rc;
        // the m panels X_j of the outer-product form of the gradient scalars, where it pays (qgd_k_dense.hip: dense_sigma_form)
        if (Np >= 128 && k.cp >= 64 && (size_t)(m - 1) * Np < (size_t)(m + 1) * k.cp && m >= 1 &&
            (!A(&k.Xouter, nt * m * panel) || !A(&k.Tlam, nt * 2 * panel) || !A(&k.Xfrag, nt * m * panel))) return rc;
        if (!getenv("QGD_BINV_OFF") && !A(&k.binv, qgdk_dense_inverse_words(k.Np, (int)nt))) return rc;      // block Gauss-Jordan inverse
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

// (QGD_CREATE_DEFER_GRID) entry points that read or size anything by the time grid allocate it first
#define NEED_GRID(h)                                                                           \
    do { if (!(h)->grid_ready) { int rg__ = alloc_grid(h); if (rg__) return rg__; } } while (0)

#define K_TRY(h, expr)                                                                         \
    do {                                                                                       \
        int e__ = (expr);                                                                      \
        if (e__ != 0)                                                                          \
            return fail((h), QGD_ERR_NO_DEVICE, std::string(#expr) + ": " + hipGetErrorString((hipError_t)e__)); \
    } while (0)

qgd_handle_s::HostReg *find_reg(qgd_handle h, const void *p, size_t bytes)
{
    for (auto &r : h->regs)
        if ((const char *)p >= (const char *)r.host && (const char *)p + bytes <= (const char *)r.host + r.bytes) return &r;
    return nullptr;
}

int copy_side(qgd_handle h)
{
    if (!h->copy_stream) HIP_TRY(h, hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
    // (two DMA engines for a large pinned download: 1.24 -> 0.87 ms for the reference-shaped cnot3 call on a box whose single
    //  engine path was slow, gpurun_out/r4e; QGD_COPY_SPLIT=0 keeps one stream)
    if (!h->copy_stream2 && !(getenv("QGD_COPY_SPLIT") && atoi(getenv("QGD_COPY_SPLIT")) == 0)) HIP_TRY(h, hipStreamCreateWithFlags(&h->copy_stream2, hipStreamNonBlocking));
    if (!h->ev_ready) HIP_TRY(h, hipEventCreateWithFlags(&h->ev_ready, hipEventDisableTiming));
    return QGD_OK;
}

// the copy stream takes over from the compute stream at this point of the launch sequence
int hand_over(qgd_handle h)
{
    HIP_TRY(h, hipEventRecord(h->ev_ready, h->k.stream));
    HIP_TRY(h, hipStreamWaitEvent(h->copy_stream, h->ev_ready, 0));
    if (h->copy_stream2) HIP_TRY(h, hipStreamWaitEvent(h->copy_stream2, h->ev_ready, 0));
    h->copies_pending = true;
    return QGD_OK;
}

int finish_copies(qgd_handle h)
{
    if (h->copies_pending) {
        // (spinning on hipStreamQuery, or on an event recorded behind the copies: no difference, 0.94 ms either way)
        HIP_TRY(h, hipStreamSynchronize(h->copy_stream));
        if (h->copy_stream2) HIP_TRY(h, hipStreamSynchronize(h->copy_stream2));
        h->copies_pending = false;
    }
    return QGD_OK;
}

// Device-to-host download on the copy stream.  A plain hipMemcpyAsync into REGISTERED host memory runs as a blit kernel
// (__amd_rocclr_copyBuffer): its waves fill the CUs and starve the adjoint chain kernels beside it (15 -> 410 us for the
// first of them on cnot3), which delays lambda and leaves the PCIe link idle at the end of the evaluation.  The same
// bytes as a pitched (rows x row_bytes, pitch = row_bytes) copy go through the DMA engine and leave the CUs alone.
// QGD_COPY_BLIT=1 keeps the plain copy.
int download(qgd_handle h, void *dst, const void *src, size_t row_bytes, size_t rows)
{
    static const bool blit = getenv("QGD_COPY_BLIT") != nullptr;
    if (blit || rows <= 1 || !find_reg(h, dst, row_bytes * rows))
        HIP_TRY(h, hipMemcpyAsync(dst, src, row_bytes * rows, hipMemcpyDeviceToHost, h->copy_stream));
    else if (h->copy_stream2 && rows >= 2 && row_bytes * rows > ((size_t)8 << 20)) {      // (experiment: two DMA engines side by side)
        const size_t r1 = rows / 2;
        HIP_TRY(h, hipMemcpy2DAsync(dst, row_bytes, src, row_bytes, row_bytes, r1, hipMemcpyDeviceToHost, h->copy_stream));
        HIP_TRY(h, hipMemcpy2DAsync((char *)dst + r1 * row_bytes, row_bytes, (const char *)src + r1 * row_bytes, row_bytes, row_bytes, rows - r1,
                                    hipMemcpyDeviceToHost, h->copy_stream2));
    } else
        HIP_TRY(h, hipMemcpy2DAsync(dst, row_bytes, src, row_bytes, row_bytes, rows, hipMemcpyDeviceToHost, h->copy_stream));
    return QGD_OK;
}

// state history: panels hist [nt][Np][2cp] (j = 0) and dpsi [nt][m][Np][2cp] (j = 1..m) -> the reference's
// uv_history[2N, 1+m, nt, c] (forward_evolution.jl:42-44).  Asynchronous: finish_copies() before returning.
// QGD_ZEROCOPY_WGS=n: n persistent workgroups of k_layout write registered (mapped) host buffers in place instead of
// staging on the device + copying.  Measured on cnot3 with all three outputs (31.6 MB): staged 0.911 ms per evaluation,
// zero-copy 1.41 / 1.15 / 1.01 / 0.96 / 0.96 ms with 8 / 16 / 32 / 64 / 128 workgroups -- the copies win, so 0 is the default.
int zerocopy_wgs()
{
    const char *e = getenv("QGD_ZEROCOPY_WGS");
    return e ? atoi(e) : 0;
}

int copy_history_out(qgd_handle h, double *uv_history, int save = 1)
{
    qgdk_ctx &k = h->k;
    // save > 1 (eval_forward's saveEveryNsteps, forward_evolution.jl:104,178,239-241): slot s of the output holds time
    // point s * save -- the re-layout kernel reads the panels with a stride of `save` time points
    const size_t hstep = (size_t)k.Np * 2 * k.cp * (size_t)save, nt = 1 + ((size_t)k.nt - 1) / (size_t)save, m = k.m, n2 = 2 * (size_t)k.N;
    const size_t total = n2 * (m + 1) * nt * k.c;
    int rc = copy_side(h);
    if (rc) return rc;
    const long long dcol = (long long)(nt * (m + 1) * n2), dn = (long long)((m + 1) * n2), dj = (long long)n2;
    qgd_handle_s::HostReg *reg = find_reg(h, uv_history, total * sizeof(double));
    if (reg && reg->dev && zerocopy_wgs() > 0) {      // registered: written in place by a few persistent workgroups on the copy stream
        double *dst = reinterpret_cast<double *>(static_cast<char *>(reg->dev) + (reinterpret_cast<char *>(uv_history) - static_cast<char *>(reg->host)));
        if ((rc = hand_over(h))) return rc;
        K_TRY(h, qgdk_layout(&k, k.hist, (long long)hstep, 0, dst, dcol, dn, dj, 0, (int)nt, 1, 0, h->copy_stream, zerocopy_wgs()));
        K_TRY(h, qgdk_layout(&k, k.dpsi, (long long)(m * hstep), (long long)(hstep / save), dst + n2, dcol, dn, dj, 0, (int)nt, (int)m, 0, h->copy_stream, zerocopy_wgs()));
        return QGD_OK;
    }
    // (the staging buffer is sized for the full grid: a strided call uses its front)
    if (!h->stage_hist && (rc = dev_alloc(h, h->stage_bufs, &h->stage_hist, n2 * (m + 1) * (size_t)k.nt * k.c))) return rc;
    K_TRY(h, qgdk_layout(&k, k.hist, (long long)hstep, 0, h->stage_hist, dcol, dn, dj, 0, (int)nt, 1, 0, k.stream, 0));
    K_TRY(h, qgdk_layout(&k, k.dpsi, (long long)(m * hstep), (long long)(hstep / save), h->stage_hist + n2, dcol, dn, dj, 0, (int)nt, (int)m, 0, k.stream, 0));
    if ((rc = hand_over(h))) return rc;
    return download(h, uv_history, h->stage_hist, n2 * (m + 1) * sizeof(double), nt * k.c);
}

// one panel per time point (lambda, adjoint forcing) -> [2N, J, nt, c] with only Taylor index 0 written
// (J = 1: adjoint_forcing; J = 1+m: lambda_history, whose other columns are zero)
int copy_panels_out(qgd_handle h, const double *panels, double **stage, double *out, size_t J, int n_first)
{
    qgdk_ctx &k = h->k;
    const size_t hstep = (size_t)k.Np * 2 * k.cp, nt = k.nt, n2 = 2 * (size_t)k.N;
    const size_t compact = n2 * nt * k.c;
    int rc = copy_side(h);
    if (rc) return rc;
    qgd_handle_s::HostReg *reg = find_reg(h, out, compact * J * sizeof(double));
    if (reg && reg->dev && zerocopy_wgs() > 0) {      // registered: the j = 0 columns are written in place; the rest is zero-filled once
        if (!reg->zeroed) { memset(out, 0, compact * J * sizeof(double)); reg->zeroed = true; }
        double *dst = reinterpret_cast<double *>(static_cast<char *>(reg->dev) + (reinterpret_cast<char *>(out) - static_cast<char *>(reg->host)));
        if ((rc = hand_over(h))) return rc;
        K_TRY(h, qgdk_layout(&k, panels, (long long)hstep, 0, dst, (long long)(nt * J * n2), (long long)(J * n2), 0, n_first, (int)nt - n_first, 1, 0,
                             h->copy_stream, zerocopy_wgs()));
        return QGD_OK;
    }
    if (!*stage) {
        if ((rc = dev_alloc(h, h->stage_bufs, stage, compact))) return rc;
        HIP_TRY(h, hipMemsetAsync(*stage, 0, compact * sizeof(double), k.stream));     // time points below n_first stay zero
    }
    K_TRY(h, qgdk_layout(&k, panels, (long long)hstep, 0, *stage, (long long)(nt * n2), (long long)n2, 0, n_first, (int)nt - n_first, 1, 0, k.stream, 0));
    if ((rc = hand_over(h))) return rc;
    if (J == 1) {
        return download(h, out, *stage, n2 * sizeof(double), nt * k.c);
    }
    if (reg) {      // pinned destination: strided copy of the j = 0 columns; the rest is zero-filled once
        if (!reg->zeroed) { memset(out, 0, compact * J * sizeof(double)); reg->zeroed = true; }
        HIP_TRY(h, hipMemcpy2DAsync(out, J * n2 * sizeof(double), *stage, n2 * sizeof(double), n2 * sizeof(double), nt * k.c,
                                    hipMemcpyDeviceToHost, h->copy_stream));
        return QGD_OK;
    }
    h->scatter_tmp.resize(compact);
    HIP_TRY(h, hipMemcpyAsync(h->scatter_tmp.data(), *stage, compact * sizeof(double), hipMemcpyDeviceToHost, h->copy_stream));
    HIP_TRY(h, hipStreamSynchronize(h->copy_stream));
    memset(out, 0, compact * J * sizeof(double));
    for (size_t r = 0; r < nt * (size_t)k.c; r++) memcpy(out + r * J * n2, h->scatter_tmp.data() + r * n2, n2 * sizeof(double));
    return QGD_OK;
}

// lambda_history with its derivative columns (qgd_set_lambda_derivatives): lam [nt][Np][2cp] (j = 0) and
// dlam [nt][m][Np][2cp] (k_adjoint_derivs, j = 1..m) -> [2N, 1+m, nt, c] for time indices 1 .. nt-1; index 0 stays
// zero, as in the reference (forward_evolution.jl:414-480).  Asynchronous: finish_copies() before returning.
int copy_lambda_full_out(qgd_handle h, double *out)
{
    qgdk_ctx &k = h->k;
    const size_t hstep = (size_t)k.Np * 2 * k.cp, nt = k.nt, m = k.m, n2 = 2 * (size_t)k.N;
    const size_t total = n2 * (m + 1) * nt * k.c;
    int rc = copy_side(h);
    if (rc) return rc;
    if (!h->dlam) {
        if ((rc = dev_alloc(h, h->stage_bufs, &h->dlam, nt * std::max<size_t>(m, 1) * hstep))) return rc;
        if ((m + 1) * (size_t)k.Np * 16 * sizeof(double) > 150 * 1024 &&
            (rc = dev_alloc(h, h->stage_bufs, &h->dlam_scratch, (nt - 1) * (size_t)(k.cp / 8) * (m + 1) * k.Np * 16))) return rc;
        if ((rc = dev_alloc(h, h->stage_bufs, &h->stage_lam_full, total))) return rc;
        HIP_TRY(h, hipMemsetAsync(h->stage_lam_full, 0, total * sizeof(double), k.stream));
    }
    { PhaseTimer t(h, "lambda_derivs"); K_TRY(h, qgdk_adjoint_derivs(&k, h->dlam, h->dlam_scratch)); }
    const long long dcol = (long long)(nt * (m + 1) * n2), dn = (long long)((m + 1) * n2), dj = (long long)n2;
    K_TRY(h, qgdk_layout(&k, k.lam, (long long)hstep, 0, h->stage_lam_full, dcol, dn, dj, 1, (int)nt - 1, 1, 0, k.stream, 0));
    K_TRY(h, qgdk_layout(&k, h->dlam, (long long)(m * hstep), (long long)hstep, h->stage_lam_full + n2, dcol, dn, dj, 1, (int)nt - 1, (int)m, 0, k.stream, 0));
    if ((rc = hand_over(h))) return rc;
    return download(h, out, h->stage_lam_full, n2 * (m + 1) * sizeof(double), nt * k.c);
}

int upload_pcof(qgd_handle h, const double *pcof, int n_pcof)
{
    if (n_pcof != h->k.n_pcof) return fail(h, QGD_ERR_ARGUMENT, "length of pcof does not match the control basis");
    const double *src = pcof;
    if (h->host_in && h->host_out_len >= (size_t)n_pcof) {   // every evaluation ends with a stream synchronisation: the buffer is free
        memcpy(h->host_in, pcof, sizeof(double) * n_pcof);
        src = h->host_in;
    }
    HIP_TRY(h, hipMemcpyAsync(h->pcof_dev, src, sizeof(double) * n_pcof, hipMemcpyHostToDevice, h->k.stream));
    return QGD_OK;
}

// forward, part 1: everything that needs no other rank (tables .. block propagators)
int forward_begin(qgd_handle h, const double *pcof, int n_pcof)
{
    qgdk_ctx &k = h->k;
    // guard penalty: the guard stage stores its workgroups' partial sums and a later stage adds them in a fixed order (the
    // same bits on every run)
    k.gpart_on = (k.gpart && !getenv("QGD_GUARD_ATOMIC")) ? 1 : 0;
    k.gpart_n = qgdk_guard_parts(&k);
    // who adds the partials up: the terminal stage where this handle runs one right behind the guard stage (the grid
    // resident and the final time its own); a window of a long grid and the other ranks of a partition launch k_guard_fold
    k.gpart_terminal = (h->chunks_eff == 1 && k.part_rank == k.part_world - 1) ? 1 : 0;
    if (pcof) {
        if (!h->have_basis) return fail(h, QGD_ERR_STATE, "qgd_set_control_basis must be called before passing pcof");
        PhaseTimer t(h, "tables");
        if (n_pcof == k.n_pcof && n_pcof <= QGD_PCOF_KERNARG && h->graph_off && !getenv("QGD_PCOF_COPY")) {   // (a captured graph would freeze the values)
            K_TRY(h, qgdk_tables_kernarg(&k, pcof, n_pcof));      // pcof rides in the kernel arguments: no copy packet
        } else {
            int rc = upload_pcof(h, pcof, n_pcof);
            if (rc) return rc;
            K_TRY(h, qgdk_tables(&k, h->pcof_dev));
        }
    } else if (!h->have_tables && k.n_ops > 0) {
        return fail(h, QGD_ERR_STATE, "no control tables: call qgd_set_control_tables or pass pcof");
    }
    if (!pcof) {   // (with pcof, k_tables clears them)
        if (!k.keep_scal) HIP_TRY(h, hipMemsetAsync(k.scal, 0, 4 * sizeof(double), k.stream));      // (a later window of a long grid keeps the running guard sum)
        HIP_TRY(h, hipMemsetAsync(k.status, 0, 2 * sizeof(int), k.stream));
    }
    const int C = std::min<int>(h->pipe_chunks, qgd_handle_s::MAX_CHUNKS);
    const bool piped = C > 1 && k.Np == 64 && k.use_sparse && qgdk_propagator_is_fused(&k) && k.scan_blocks >= 2 * C &&
                       !(h->timing && h->timing_only.empty());      // (a full per-phase breakdown is taken on the serial path)
    if (piped) {
        const int B = k.scan_blocks, S = k.nt - 1;
        const size_t pl = (size_t)k.Np * k.Np, panel = 2 * pl;
        for (int i = 0; i < C; i++) {
            if (!h->pipe_stream[i]) {
                HIP_TRY(h, hipStreamCreateWithFlags(&h->pipe_stream[i], hipStreamNonBlocking));
                HIP_TRY(h, hipEventCreateWithFlags(&h->pipe_built[i], hipEventDisableTiming));
                HIP_TRY(h, hipEventCreateWithFlags(&h->pipe_done[i], hipEventDisableTiming));
            }
            const int b0 = (int)((long long)B * i / C), b1 = (int)((long long)B * (i + 1) / C);
            const int s_lo = std::min(S, b0 * k.scan_blen), s_hi = std::min(S, b1 * k.scan_blen);
            if (s_hi <= s_lo) continue;
            // build: time points (s_lo, s_hi] (and point 0 with the first chunk).  The launchers index everything by the
            // time point: a context whose per-time-point arrays start at n0 and whose grid has `cnt` points does the piece.
            {
                const int n0 = (i == 0) ? 0 : s_lo + 1, cnt = s_hi - n0 + 1;
                qgdk_ctx kb = k;
                kb.tab += (size_t)n0 * (k.m + 1) * std::max(k.n_ops, 1) * 2; kb.L += (size_t)n0 * panel; kb.R += (size_t)n0 * panel;
                kb.nt = cnt;
                PhaseTimer t(h, "build_LR", k.stream, i);
                K_TRY(h, qgdk_build_LR(&kb));
            }
            HIP_TRY(h, hipEventRecord(h->pipe_built[i], k.stream));
            HIP_TRY(h, hipStreamWaitEvent(h->pipe_stream[i], h->pipe_built[i], 0));
            // inverse + propagator of the matrices n in (s_lo, s_hi]: L_n^-1 and P_{n-1} = L_n^-1 R_{n-1}
            {
                qgdk_ctx ki = k;
                ki.stream = h->pipe_stream[i];
                ki.L += (size_t)s_lo * panel; ki.R += (size_t)s_lo * panel; ki.LinvT += (size_t)s_lo * 2 * pl;
                ki.LinvA += (size_t)s_lo * 2 * pl; ki.Pr += (size_t)s_lo * panel; ki.Pc += (size_t)s_lo * 2 * pl;
                ki.nt = s_hi - s_lo + 1;
                PhaseTimer t(h, "inverse", ki.stream, i);
                K_TRY(h, qgdk_inverse(&ki));
            }
            { PhaseTimer t(h, "sweep_forward", h->pipe_stream[i], i); K_TRY(h, qgdk_forward_blocks_range(&k, b0, b1, h->pipe_stream[i])); }
            HIP_TRY(h, hipEventRecord(h->pipe_done[i], h->pipe_stream[i]));
        }
        for (int i = 0; i < C; i++) HIP_TRY(h, hipStreamWaitEvent(k.stream, h->pipe_done[i], 0));
        { PhaseTimer t(h, "sweep_forward", k.stream, C); K_TRY(h, qgdk_forward_blocks_upper(&k)); }
    } else {
        { PhaseTimer t(h, "build_LR"); K_TRY(h, qgdk_build_LR(&k)); }
        { PhaseTimer t(h, "inverse"); K_TRY(h, qgdk_inverse(&k)); }
        if (qgdk_propagator_is_fused(&k)) { K_TRY(h, qgdk_propagator(&k)); }   // k_inverse_mfma formed P_n already
        else { PhaseTimer t(h, "propagator"); K_TRY(h, qgdk_propagator(&k)); }
        { PhaseTimer t(h, "sweep_forward"); K_TRY(h, qgdk_forward_blocks(&k)); }
    }
    h->forward_valid = false;
    h->derivs_valid = false;
    return QGD_OK;
}

// forward, part 2: after the block propagators of all ranks are in PiX
int forward_end(qgd_handle h)
{
    qgdk_ctx &k = h->k;
    { PhaseTimer t(h, "sweep_forward2"); K_TRY(h, qgdk_forward_finish(&k)); }
    if (k.have_guard == 0 && h->forcing_zero) {
        // nothing to do: without a guard projector the kernel only re-clears the forcing (6 us of the 100 us of a cnot2 evaluation)
    } else if (!qgdk_guard_is_fused(&k)) {
        PhaseTimer t(h, "guard"); K_TRY(h, qgdk_guard(&k));      // else: done by the history pass
        if (k.have_guard == 0) h->forcing_zero = true;
    }
    if (k.part_rank == k.part_world - 1 && !h->defer_terminal) {   // the rank that owns the final time
        PhaseTimer t(h, "terminal"); K_TRY(h, qgdk_terminal(&k, k.have_target));
    }
    if (k.gpart_on && !k.gpart_terminal && k.have_guard) { PhaseTimer t(h, "guard"); K_TRY(h, qgdk_guard_fold(&k)); }
    h->forward_valid = true;
    return QGD_OK;
}

int adjoint_begin(qgd_handle h)
{
    qgdk_ctx &k = h->k;
    k.fuse_terminal = h->defer_terminal ? 1 : 0;     // (the forward sweep left the terminal condition to this launch)
    h->defer_terminal = false;
    { PhaseTimer t(h, "sweep_adjoint"); K_TRY(h, qgdk_adjoint_blocks(&k)); }
    k.fuse_terminal = 0;
    return QGD_OK;
}

int adjoint_end(qgd_handle h)
{
    qgdk_ctx &k = h->k;
    { PhaseTimer t(h, "sweep_adjoint2"); K_TRY(h, qgdk_adjoint_finish(&k)); }
    { PhaseTimer t(h, "lambda"); K_TRY(h, qgdk_lambda(&k)); }
    if (h->lambda_out) {      // its download runs beside the gradient kernels
        double *out = h->lambda_out; h->lambda_out = nullptr;
        int rc = h->lambda_derivs ? copy_lambda_full_out(h, out) : copy_panels_out(h, k.lam, &h->stage_lam, out, (size_t)k.m + 1, 1);
        if (rc) return rc;
    }
    if (!h->derivs_valid && qgdk_gradient_needs_derivs(&k)) { PhaseTimer t(h, "derivs"); K_TRY(h, qgdk_derivs(&k)); h->derivs_valid = true; }
    { PhaseTimer t(h, "gradient"); K_TRY(h, qgdk_gradient(&k)); }
    return QGD_OK;
}

// ---------------------------------------------------------------------------
// Bounded-memory time grid.  The per-time-point matrices (D, L, R, L^-1, P: ~330 KB per step at cnot3, 18 MB at N = 256)
// of a long grid do not have to be resident together: the grid is cut into chunks_eff windows that use the SAME
// buffers one after the other.  Forward pass, windows in order: build -> inverse -> block products -> history of the
// window from the state the previous window ended in; only that state (one panel per window) is kept.  Adjoint pass,
// windows in reverse: the window's matrices and forward history are formed again from its stored start state (unless
// they are the ones still in the buffers), then the adjoint scan of the window from the y the next window ended in,
// lambda, and the window's share of the gradient, which k_contract ADDS to grad.  Cost: build + inverse + forward
// history once more for all windows but one.  The reference keeps O(nsteps) state history but no matrices at all
// (matrix-free GMRES); its low-order runs with 10^4 .. 10^6 steps (examples/cnot3_optimize_gate.sb:27-40) are what
// this mode is for.
// ---------------------------------------------------------------------------
// The reference-layout outputs of a chunked grid, one window at a time: the window's panels are re-laid out into a compact
// staging buffer on the device and copied into the caller's FULL array at the window's time offset (a pitched copy per
// column); the copy is awaited before the next window overwrites the panels.  Windows share their end points (same values).
int window_history_out(qgd_handle h, double *uv_history, int save = 1)       // [2N, 1+m, 1 + (nt_glob-1)/save, c]
{
    qgdk_ctx &k = h->k;
    const size_t hstep = (size_t)k.Np * 2 * k.cp, m = k.m, n2 = 2 * (size_t)k.N, ntg = k.nt_glob;
    const size_t nt0 = std::min<size_t>((size_t)k.bpr * k.scan_blen + 1, ntg);      // the longest window
    int rc = copy_side(h);
    if (rc) return rc;
    if (!h->stage_hist && (rc = dev_alloc(h, h->stage_bufs, &h->stage_hist, n2 * (m + 1) * nt0 * k.c))) return rc;
    // saveEveryNsteps (forward_evolution.jl:104,178,239-241): slot s of the output holds GLOBAL time point s * save; this
    // window holds the global points n_off .. n_off + nt - 1 (windows share their end points: same values, same slot)
    const size_t sv = (size_t)save, g_lo = (size_t)k.n_off, g_hi = (size_t)k.n_off + (size_t)k.nt - 1;
    const size_t s_lo = (g_lo + sv - 1) / sv, s_hi = g_hi / sv;
    if (s_hi < s_lo) return QGD_OK;                                                  // (no saved point falls into this window)
    const size_t cnt = s_hi - s_lo + 1, loc = s_lo * sv - g_lo, slots = 1 + (ntg - 1) / sv;
    const long long dcol = (long long)(cnt * (m + 1) * n2), dn = (long long)((m + 1) * n2), dj = (long long)n2;
    K_TRY(h, qgdk_layout(&k, k.hist + loc * hstep, (long long)(hstep * sv), 0, h->stage_hist, dcol, dn, dj, 0, (int)cnt, 1, 0, k.stream, 0));
    K_TRY(h, qgdk_layout(&k, k.dpsi + loc * m * hstep, (long long)(m * hstep * sv), (long long)hstep, h->stage_hist + n2, dcol, dn, dj, 0, (int)cnt, (int)m, 0, k.stream, 0));
    if ((rc = hand_over(h))) return rc;
    const size_t row = cnt * (m + 1) * n2 * sizeof(double);
    HIP_TRY(h, hipMemcpy2DAsync(uv_history + s_lo * (m + 1) * n2, slots * (m + 1) * n2 * sizeof(double), h->stage_hist, row, row,
                                (size_t)k.c, hipMemcpyDeviceToHost, h->copy_stream));
    return finish_copies(h);
}

// lambda_history of a window WITH its derivative columns (qgd_set_lambda_derivatives): local time indices 1 .. nt-1 (the
// window's first point is the previous window's last; global index 0 is never written, as in the reference)
int window_lambda_full_out(qgd_handle h, double *out)       // [2N, 1+m, nt_glob, c]
{
    qgdk_ctx &k = h->k;
    const size_t hstep = (size_t)k.Np * 2 * k.cp, nt = k.nt, m = k.m, n2 = 2 * (size_t)k.N, ntg = k.nt_glob;
    const size_t nt0 = std::min<size_t>((size_t)k.bpr * k.scan_blen + 1, ntg);
    int rc = copy_side(h);
    if (rc) return rc;
    if (nt < 2) return QGD_OK;
    if (!h->dlam) {
        if ((rc = dev_alloc(h, h->stage_bufs, &h->dlam, nt0 * std::max<size_t>(m, 1) * hstep))) return rc;
        if ((m + 1) * (size_t)k.Np * 16 * sizeof(double) > 150 * 1024 &&
            (rc = dev_alloc(h, h->stage_bufs, &h->dlam_scratch, (nt0 - 1) * (size_t)(k.cp / 8) * (m + 1) * k.Np * 16))) return rc;
        if ((rc = dev_alloc(h, h->stage_bufs, &h->stage_lam_full, n2 * (m + 1) * nt0 * k.c))) return rc;
    }
    { PhaseTimer t(h, "lambda_derivs"); K_TRY(h, qgdk_adjoint_derivs(&k, h->dlam, h->dlam_scratch)); }
    const long long dcol = (long long)(nt * (m + 1) * n2), dn = (long long)((m + 1) * n2), dj = (long long)n2;
    K_TRY(h, qgdk_layout(&k, k.lam, (long long)hstep, 0, h->stage_lam_full, dcol, dn, dj, 1, (int)nt - 1, 1, 0, k.stream, 0));
    K_TRY(h, qgdk_layout(&k, h->dlam, (long long)(m * hstep), (long long)hstep, h->stage_lam_full + n2, dcol, dn, dj, 1, (int)nt - 1, (int)m, 0, k.stream, 0));
    if ((rc = hand_over(h))) return rc;
    HIP_TRY(h, hipMemcpy2DAsync(out + ((size_t)k.n_off + 1) * (m + 1) * n2, ntg * (m + 1) * n2 * sizeof(double),
                                h->stage_lam_full + (m + 1) * n2, nt * (m + 1) * n2 * sizeof(double), (nt - 1) * (m + 1) * n2 * sizeof(double),
                                (size_t)k.c, hipMemcpyDeviceToHost, h->copy_stream));
    return finish_copies(h);
}

int window_panels_out(qgd_handle h, const double *panels, double **stage, double *out, size_t J, int n_first)     // [2N, J, nt_glob, c], j = 0
{
    qgdk_ctx &k = h->k;
    const size_t hstep = (size_t)k.Np * 2 * k.cp, nt = k.nt, n2 = 2 * (size_t)k.N, ntg = k.nt_glob;
    const size_t nt0 = std::min<size_t>((size_t)k.bpr * k.scan_blen + 1, ntg);
    int rc = copy_side(h);
    if (rc) return rc;
    if (!*stage && (rc = dev_alloc(h, h->stage_bufs, stage, n2 * nt0 * k.c))) return rc;
    K_TRY(h, qgdk_layout(&k, panels, (long long)hstep, 0, *stage, (long long)(nt * n2), (long long)n2, 0, n_first, (int)nt - n_first, 1, 0, k.stream, 0));
    if ((rc = hand_over(h))) return rc;
    const size_t cnt = nt - (size_t)n_first;
    if (cnt) {
        if (J == 1) {
            HIP_TRY(h, hipMemcpy2DAsync(out + ((size_t)k.n_off + n_first) * n2, ntg * n2 * sizeof(double), *stage + (size_t)n_first * n2,
                                        nt * n2 * sizeof(double), cnt * n2 * sizeof(double), (size_t)k.c, hipMemcpyDeviceToHost, h->copy_stream));
        } else {
            for (size_t col = 0; col < (size_t)k.c; col++)      // rows of 2N doubles, J * 2N apart in the caller's array
                HIP_TRY(h, hipMemcpy2DAsync(out + ((col * ntg + k.n_off + n_first) * J) * n2, J * n2 * sizeof(double),
                                            *stage + (col * nt + n_first) * n2, n2 * sizeof(double), n2 * sizeof(double), cnt,
                                            hipMemcpyDeviceToHost, h->copy_stream));
        }
    }
    return finish_copies(h);
}

// qgd_set_control_tables on a windowed grid: the window's slice of the caller's tables goes to the device before the
// window's matrices are built (with pcof the tables kernel forms them from the basis, which covers the whole grid)
int window_tables(qgd_handle h)
{
    qgdk_ctx &k = h->k;
    const size_t per = (size_t)(k.m + 1) * k.n_ops, cnt = (size_t)k.nt * per, off = (size_t)k.n_off * per;
    if (h->tab_p_host.size() < off + cnt) return fail(h, QGD_ERR_STATE, "no control tables: call qgd_set_control_tables or pass pcof");
    double *tmp = nullptr;
    HIP_TRY(h, hipMalloc((void **)&tmp, 2 * cnt * sizeof(double) + 64));
    hipError_t e1 = hipMemcpyAsync(tmp, h->tab_p_host.data() + off, cnt * sizeof(double), hipMemcpyHostToDevice, k.stream);
    hipError_t e2 = hipMemcpyAsync(tmp + cnt, h->tab_q_host.data() + off, cnt * sizeof(double), hipMemcpyHostToDevice, k.stream);
    int kr = (e1 == hipSuccess && e2 == hipSuccess) ? qgdk_tables_from_host(&k, tmp, tmp + cnt) : 1;
    (void)hipStreamSynchronize(k.stream);
    (void)hipFree(tmp);
    if (kr) return fail(h, QGD_ERR_NO_DEVICE, "uploading control tables failed");
    return QGD_OK;
}

int chunk_forward(qgd_handle h, const double *pcof, int n_pcof, int r, bool rerun)
{
    qgdk_ctx &k = h->k;
    int rc = plan_windows(h, h->chunks_req, r);
    if (rc) return rc;
    if (!pcof && k.n_ops > 0 && (rc = window_tables(h))) return rc;
    const size_t hstep = (size_t)k.Np * 2 * k.cp;
    const double *start = h->chunk_state + (size_t)r * hstep;
    for (double *dst : {k.psi0, k.hist, k.bnd, k.bnd2})
        HIP_TRY(h, hipMemcpyAsync(dst, start, hstep * sizeof(double), hipMemcpyDeviceToDevice, k.stream));
    k.keep_scal = (r > 0 || rerun) ? 1 : 0;
    double *scal_real = k.scal;
    if (rerun) k.scal = h->scal_scratch;          // (the guard sum of this window was counted by the forward pass)
    rc = forward_begin(h, pcof, n_pcof);
    if (!rc) { PhaseTimer t(h, "sweep_forward2"); int e = qgdk_forward_finish(&k); if (e) rc = fail(h, QGD_ERR_NO_DEVICE, "forward history pass failed to launch"); }
    if (!rc && !qgdk_guard_is_fused(&k) && (k.have_guard || !h->forcing_zero)) {
        PhaseTimer t(h, "guard");
        if (qgdk_guard(&k)) rc = fail(h, QGD_ERR_NO_DEVICE, "guard kernel failed to launch");
        if (k.have_guard == 0) h->forcing_zero = true;
    }
    if (!rc && k.gpart_on && k.have_guard && qgdk_guard_fold(&k)) rc = fail(h, QGD_ERR_NO_DEVICE, "guard fold failed to launch");
    k.scal = scal_real; k.keep_scal = 0;
    if (rc) return rc;
    if (!rerun)
        HIP_TRY(h, hipMemcpyAsync(h->chunk_state + (size_t)(r + 1) * hstep, k.hist + (size_t)(k.nt - 1) * hstep, hstep * sizeof(double),
                                  hipMemcpyDeviceToDevice, k.stream));
    h->resident_window = r;
    return QGD_OK;
}

int chunked_forward(qgd_handle h, const double *pcof, int n_pcof, double *uv_history = nullptr, int save = 1)
{
    qgdk_ctx &k = h->k;
    if (!pcof && !h->have_tables && k.n_ops > 0) return fail(h, QGD_ERR_STATE, "no control tables: call qgd_set_control_tables or pass pcof");
    int rc;
    for (int r = 0; r < h->chunks_eff; r++) {
        if ((rc = chunk_forward(h, pcof, n_pcof, r, false))) return rc;
        if (uv_history) {      // the window's share of the state history with its stage derivatives
            { PhaseTimer t(h, "derivs"); K_TRY(h, qgdk_derivs(&k)); }
            if ((rc = window_history_out(h, uv_history, save))) return rc;
        }
    }
    { PhaseTimer t(h, "terminal"); K_TRY(h, qgdk_terminal(&k, k.have_target)); }      // overlaps (and y_N) from the final state
    h->forward_valid = true; h->derivs_valid = false;
    if (pcof) h->fwd_pcof.assign(pcof, pcof + n_pcof); else h->fwd_pcof.clear();
    return QGD_OK;
}

int chunked_adjoint(qgd_handle h, double *lambda_history = nullptr, double *adjoint_forcing = nullptr)
{
    qgdk_ctx &k = h->k;
    const size_t hstep = (size_t)k.Np * 2 * k.cp;
    const int W = h->chunks_eff;
    int rc;
    if (lambda_history)      // (the library writes the j = 0 columns; the others, and time index 0, are zero as in the resident call)
        memset(lambda_history, 0, sizeof(double) * 2 * (size_t)k.N * (k.m + 1) * (size_t)k.nt_glob * k.c);
    for (int r = W - 1; r >= 0; r--) {
        if (h->resident_window != r) {
            if ((rc = chunk_forward(h, h->fwd_pcof.empty() ? nullptr : h->fwd_pcof.data(), (int)h->fwd_pcof.size(), r, true))) return rc;
        } else if ((rc = plan_windows(h, h->chunks_req, r))) return rc;
        if (adjoint_forcing && (rc = window_panels_out(h, k.forcing, &h->stage_f, adjoint_forcing, 1, 0))) return rc;
        if (r == W - 1) { PhaseTimer t(h, "terminal"); K_TRY(h, qgdk_terminal(&k, 1)); }
        else            // y at the end of this window = y at the start of the next one
            for (double *dst : {k.yhist + (size_t)(k.nt - 1) * hstep, k.bndY + (size_t)k.scan_blocks * hstep, k.bndY2 + (size_t)k.scan_blocks2 * hstep})
                HIP_TRY(h, hipMemcpyAsync(dst, h->carry_y, hstep * sizeof(double), hipMemcpyDeviceToDevice, k.stream));
        k.grad_accumulate = (r != W - 1) ? 1 : 0;
        h->derivs_valid = false;
        rc = adjoint_begin(h);
        if (!rc) rc = adjoint_end(h);
        k.grad_accumulate = 0;
        if (rc) return rc;
        if (lambda_history && (rc = h->lambda_derivs ? window_lambda_full_out(h, lambda_history)
                                                     : window_panels_out(h, k.lam, &h->stage_lam, lambda_history, (size_t)k.m + 1, 1))) return rc;
        HIP_TRY(h, hipMemcpyAsync(h->carry_y, k.yhist, hstep * sizeof(double), hipMemcpyDeviceToDevice, k.stream));
    }
    h->forward_valid = true;          // (the window-boundary states of this pcof are still there for history_precomputed)
    return QGD_OK;
}

int check_status(qgd_handle h);
// eval_adjoint on a windowed grid: windows in reverse; each forms its matrices, takes its slice of the caller's forcing and
// the y the next window ended in (the last one: y_N = L_N^H lambda_N from the given terminal condition), runs the adjoint
// scan and lambda, and writes its share of lambda_history (global time indices n_off+1 .. n_off+nt-1).  No forward history is
// needed (forward_evolution.jl:352-483 reads none).
int chunked_eval_adjoint(qgd_handle h, const double *pcof, int n_pcof, const double *terminal_condition, const double *forcing,
                         double *lambda_history)
{
    qgdk_ctx &k = h->k;
    if (!pcof && !h->have_tables && k.n_ops > 0) return fail(h, QGD_ERR_STATE, "no control tables: call qgd_set_control_tables or pass pcof");
    const size_t Np = k.Np, PWc = 2 * k.cp, hstep = Np * PWc, N = k.N, n2 = 2 * N, m = k.m, ntg = (size_t)h->nsteps + 1;
    const int W = h->chunks_eff;
    std::vector<double> lamN(hstep, 0.0), f;
    for (size_t col = 0; col < (size_t)k.c; col++)
        for (size_t i = 0; i < N; i++) {
            const size_t o = panel_index((int)i, (int)col, (int)PWc);
            lamN[o] = terminal_condition[i + n2 * col];
            lamN[o + 8] = terminal_condition[N + i + n2 * col];
        }
    memset(lambda_history, 0, sizeof(double) * n2 * (m + 1) * ntg * k.c);
    h->forward_valid = false; h->resident_window = -1;      // (the buffers will hold no window's forward history)
    int rc;
    for (int r = W - 1; r >= 0; r--) {
        if ((rc = plan_windows(h, h->chunks_req, r))) return rc;
        const size_t nt = k.nt, n_off = k.n_off;
        if (!pcof && k.n_ops > 0 && (rc = window_tables(h))) return rc;
        if ((rc = forward_begin(h, pcof, n_pcof))) return rc;
        { PhaseTimer t(h, "sweep_forward2"); K_TRY(h, qgdk_forward_finish(&k)); }      // (the super-block propagators)
        f.assign(nt * hstep, 0.0);
        if (forcing)
            for (size_t col = 0; col < (size_t)k.c; col++)
                for (size_t n = 0; n < nt; n++) {
                    const double *src = forcing + (col * ntg + n_off + n) * n2;
                    for (size_t i = 0; i < N; i++) {
                        const size_t o = n * hstep + panel_index((int)i, (int)col, (int)PWc);
                        f[o] = src[i]; f[o + 8] = src[N + i];
                    }
                }
        // (on the library's stream: behind the history pass, which writes the guard forcing of the window into this buffer)
        HIP_TRY(h, hipMemcpyAsync(k.forcing, f.data(), f.size() * sizeof(double), hipMemcpyHostToDevice, k.stream));
        HIP_TRY(h, hipStreamSynchronize(k.stream));         // (f is filled again for the next window)
        h->forcing_zero = false;
        if (r == W - 1) {
            HIP_TRY(h, hipMemcpyAsync(k.lam + (nt - 1) * hstep, lamN.data(), hstep * sizeof(double), hipMemcpyHostToDevice, k.stream));
            K_TRY(h, qgdk_apply_LH(&k));
        } else {
            for (double *dst : {k.yhist + (nt - 1) * hstep, k.bndY + (size_t)k.scan_blocks * hstep, k.bndY2 + (size_t)k.scan_blocks2 * hstep})
                HIP_TRY(h, hipMemcpyAsync(dst, h->carry_y, hstep * sizeof(double), hipMemcpyDeviceToDevice, k.stream));
        }
        if ((rc = adjoint_begin(h))) return rc;
        { PhaseTimer t(h, "sweep_adjoint2"); K_TRY(h, qgdk_adjoint_finish(&k)); }
        { PhaseTimer t(h, "lambda"); K_TRY(h, qgdk_lambda(&k)); }
        if ((rc = check_status(h))) return rc;
        if (r == W - 1)      // lambda_N is the given one (not L_N^-H L_N^H of it)
            HIP_TRY(h, hipMemcpyAsync(k.lam + (nt - 1) * hstep, lamN.data(), hstep * sizeof(double), hipMemcpyHostToDevice, k.stream));
        if ((rc = h->lambda_derivs ? window_lambda_full_out(h, lambda_history)
                                   : window_panels_out(h, k.lam, &h->stage_lam, lambda_history, m + 1, 1))) return rc;
        HIP_TRY(h, hipMemcpyAsync(h->carry_y, k.yhist, hstep * sizeof(double), hipMemcpyDeviceToDevice, k.stream));
    }
    HIP_TRY(h, hipStreamSynchronize(k.stream));
    return QGD_OK;
}

// buffers of the forced forward sweep for (up to) nt time points and B scan blocks
int forcing_buffers(qgd_handle h, size_t nt, size_t B)
{
    qgdk_ctx &k = h->k;
    const size_t m = k.m, hstep = (size_t)k.Np * 2 * k.cp;
    const size_t key = nt * 4099u + B;
    int rc;
    if (h->forcing_key != key) {
        free_pool(h->forcing_bufs); h->forcing_key = 0;
        if ((rc = dev_alloc(h, h->forcing_bufs, &k.ff_F, nt * m * hstep))) return rc;
        if ((rc = dev_alloc(h, h->forcing_bufs, &k.ff_E, nt * m * hstep))) return rc;
        if ((rc = dev_alloc(h, h->forcing_bufs, &k.ff_XR, nt * hstep))) return rc;
        if ((rc = dev_alloc(h, h->forcing_bufs, &k.ff_XL, nt * hstep))) return rc;
        if ((rc = dev_alloc(h, h->forcing_bufs, &k.ff_Q, nt * hstep))) return rc;
        if ((rc = dev_alloc(h, h->forcing_bufs, &k.ff_phi, (B + 1) * hstep))) return rc;
        if ((rc = dev_alloc(h, h->forcing_bufs, &k.ff_bnd, (B + 2) * hstep))) return rc;
        h->fsc_forcing = nullptr;     // (N > 64 at high order: the m+2 work panels of k_forcing_terms do not fit in LDS)
        if ((size_t)(m + 2) * k.Np * 16 * sizeof(double) > 150 * 1024 &&
            (rc = dev_alloc(h, h->forcing_bufs, &h->fsc_forcing, nt * (size_t)(k.cp / 8) * (m + 2) * k.Np * 16))) return rc;
        h->forcing_key = key;
    }
    k.fs_scratch = h->fsc_forcing;
    return QGD_OK;
}

// forcing [2N, m, nt_glob, c] (Julia layout, forward_evolution.jl:42-44), time points n_off .. n_off + nt - 1 -> panels [nt][m][Np][2cp]
int upload_forcing(qgd_handle h, const double *forcing, size_t nt, size_t n_off)
{
    qgdk_ctx &k = h->k;
    const size_t m = k.m, N = k.N, n2 = 2 * N, PWc = 2 * k.cp, hstep = (size_t)k.Np * PWc, ntg = (size_t)h->nsteps + 1;
    std::vector<double> f(nt * m * hstep, 0.0);
    for (size_t col = 0; col < (size_t)k.c; col++) for (size_t n = 0; n < nt; n++) for (size_t j = 0; j < m; j++) {
        const double *src = forcing + ((col * ntg + n_off + n) * m + j) * n2;
        double *dst = f.data() + (n * m + j) * hstep;
        for (size_t i = 0; i < N; i++) {
            const size_t o = panel_index((int)i, (int)col, (int)PWc);
            dst[o] = src[i]; dst[o + 8] = src[N + i];
        }
    }
    HIP_TRY(h, hipMemcpyAsync(k.ff_F, f.data(), f.size() * sizeof(double), hipMemcpyHostToDevice, k.stream));
    HIP_TRY(h, hipStreamSynchronize(k.stream));
    return QGD_OK;
}

int fetch_results(qgd_handle h, double *grad, double *out3, const double *src);

// eval_forward(...; forcing) on a windowed grid (forward_evolution.jl:118-129,167-206): windows in order, each from the
// forced state the previous one ended in, with its slice of the caller's forcing; the guard penalty accumulates over the
// windows, the overlaps come from the final state; uv_history (stage derivatives w_j = D_j w_0 + E_j included) window by
// window as in chunked_forward.
int chunked_forward_forced(qgd_handle h, const double *pcof, int n_pcof, const double *forcing, double *uv_history, double *out3)
{
    qgdk_ctx &k = h->k;
    if (!pcof && !h->have_tables && k.n_ops > 0) return fail(h, QGD_ERR_STATE, "no control tables: call qgd_set_control_tables or pass pcof");
    const size_t hstep = (size_t)k.Np * 2 * k.cp;
    int rc;
    h->forward_valid = false; h->resident_window = -1;       // (the window-boundary states are those of the FORCED sweep from here on)
    size_t nt0 = 0, B0 = 0;
    for (int r = 0; r < h->chunks_eff; r++) {
        if ((rc = plan_windows(h, h->chunks_req, r))) return rc;
        if (r == 0) { nt0 = (size_t)k.nt; B0 = (size_t)k.scan_blocks; }      // (the first window is the longest)
        if (!pcof && k.n_ops > 0 && (rc = window_tables(h))) return rc;
        const double *start = h->chunk_state + (size_t)r * hstep;
        for (double *dst : {k.psi0, k.hist, k.bnd, k.bnd2})
            HIP_TRY(h, hipMemcpyAsync(dst, start, hstep * sizeof(double), hipMemcpyDeviceToDevice, k.stream));
        k.keep_scal = (r > 0) ? 1 : 0;
        rc = forward_begin(h, pcof, n_pcof);
        if (!rc) rc = forcing_buffers(h, std::max(nt0, (size_t)k.nt), std::max(B0, (size_t)k.scan_blocks));
        if (!rc) rc = upload_forcing(h, forcing, (size_t)k.nt, (size_t)k.n_off);
        if (!rc && qgdk_forcing_terms(&k)) rc = fail(h, QGD_ERR_NO_DEVICE, "forcing terms failed to launch");
        if (!rc && qgdk_forcing_sweep(&k)) rc = fail(h, QGD_ERR_NO_DEVICE, "forced sweep failed to launch");
        if (!rc) {
            k.gpart_n = k.nt;                                 // (one partial penalty per time point: the stand-alone guard kernel)
            if (qgdk_guard_kernel(&k)) rc = fail(h, QGD_ERR_NO_DEVICE, "guard kernel failed to launch");
            if (k.have_guard == 0) h->forcing_zero = true; else h->forcing_zero = false;
        }
        if (!rc && k.gpart_on && k.have_guard && qgdk_guard_fold(&k)) rc = fail(h, QGD_ERR_NO_DEVICE, "guard fold failed to launch");
        k.keep_scal = 0;
        if (rc) return rc;
        HIP_TRY(h, hipMemcpyAsync(h->chunk_state + (size_t)(r + 1) * hstep, k.hist + (size_t)(k.nt - 1) * hstep, hstep * sizeof(double),
                                  hipMemcpyDeviceToDevice, k.stream));
        if (uv_history) {
            { PhaseTimer t(h, "derivs"); K_TRY(h, qgdk_derivs(&k)); }
            K_TRY(h, qgdk_forcing_add_derivs(&k));            // w_j = D_j w_0 + E_j
            if ((rc = window_history_out(h, uv_history, h->save_every))) return rc;
        }
    }
    { PhaseTimer t(h, "terminal"); K_TRY(h, qgdk_terminal(&k, k.have_target)); }
    h->derivs_valid = false;
    h->fwd_pcof.clear();
    if ((rc = fetch_results(h, nullptr, out3, nullptr))) { (void)finish_copies(h); return rc; }
    return finish_copies(h);
}

#define NEEDS_RESIDENT_GRID(h, what)                                                                                   \
    do { if ((h)->chunks_eff > 1) return fail((h), QGD_ERR_UNSUPPORTED, std::string(what) + " needs the whole time grid resident: this handle " \
                                               "processes it in " + std::to_string((h)->chunks_eff) + " windows (raise qgd_set_memory_budget)"); } while (0)

int run_forward(qgd_handle h, const double *pcof, int n_pcof)
{
    if (h->chunks_eff > 1) return chunked_forward(h, pcof, n_pcof);
    if (h->part_world != 1) return fail(h, QGD_ERR_STATE, "partitioned handle: use the qgd_dist_* entry points");
    int rc = forward_begin(h, pcof, n_pcof);
    if (rc) return rc;
    if ((rc = forward_end(h))) return rc;
    if (pcof) h->fwd_pcof.assign(pcof, pcof + n_pcof); else h->fwd_pcof.clear();
    h->history_stale = false;
    return QGD_OK;
}

int fetch_results(qgd_handle h, double *grad, double *out3, const double *src);

// the small-problem evaluation (qgd_k_tiny.hip); results through the mirror when there is one
bool tiny_applies(qgd_handle h, const double *pcof, int n_pcof)
{
    const qgdk_ctx &k = h->k;
    return h->small_path && pcof && h->have_basis && n_pcof == k.n_pcof && !h->timing && h->graph_off && !h->comm && h->part_world == 1 &&
           h->chunks_eff == 1 && k.redbuf && h->host_out && qgdk_tiny_supported(&k, n_pcof) != 0;
}

int tiny_evaluate(qgd_handle h, const double *pcof, int n_pcof, bool gradient, double *grad, double *out3)
{
    qgdk_ctx &k = h->k;
    const bool mirror = h->mirror_dev != nullptr && h->mirror_ticket != nullptr;
    if (mirror) { k.mirror_dev = h->mirror_dev; k.mirror_ticket = h->mirror_ticket; k.mirror_seq = ++h->mirror_seq; }
    // (the status word: the general path clears it in its first kernel and sets it in a later one; here the kernel that could
    //  clear it -- the per-time-point front -- is also the one that sets it, so it is cleared from the host, and only after an
    //  evaluation that left it set)
    if (h->status_dirty) { HIP_TRY(h, hipMemsetAsync(k.status, 0, 2 * sizeof(int), k.stream)); h->status_dirty = false; }
    int e = qgdk_tiny_eval(&k, pcof, n_pcof, gradient ? 1 : 0);
    if (!e && gradient) e = qgdk_contract_rows(&k, k.nt);
    k.mirror_dev = nullptr;
    if (e) return fail(h, QGD_ERR_NO_DEVICE, std::string("small-problem evaluation failed to launch: ") + hipGetErrorString((hipError_t)e));
    h->mirror_armed = mirror;
    h->forward_valid = true; h->derivs_valid = false; h->history_stale = true; h->forcing_zero = false;
    h->fwd_pcof.clear();
    h->tiny_pcof.assign(pcof, pcof + n_pcof); h->tiny_was_gradient = gradient;
    return fetch_results(h, gradient ? grad : nullptr, out3, nullptr);
}

int check_status(qgd_handle h)
{
    int st = 0;
    HIP_TRY(h, hipMemcpyAsync(&st, h->k.status, sizeof(int), hipMemcpyDeviceToHost, h->k.stream));
    HIP_TRY(h, hipStreamSynchronize(h->k.stream));
    if (st) return fail(h, QGD_ERR_NUMERIC, "singular implicit step matrix L(t_n)");
    return QGD_OK;
}

int comm_wait(qgd_handle h);      // (bounded wait of a collective evaluation, defined with the RCCL binding below)

// status + results of an evaluation in one device-to-host copy (the three separate copies cost
// ~25 us of the 0.5 ms evaluation on cnot3)
int fetch_results(qgd_handle h, double *grad, double *out3, const double *src = nullptr)
{
    qgdk_ctx &k = h->k;
    if (!src) src = k.redbuf;
    if (!k.redbuf || !h->host_out) {
        int rc = check_status(h);
        if (rc) return rc;
        // (src != redbuf: the reduced [grad | scalars | flag] of a time-sharded collective evaluation -- never the rank's own sums)
        const double *g = (src && src != k.redbuf) ? src : k.grad, *sc = (src && src != k.redbuf) ? src + k.n_pcof : k.scal;
        if (src && src != k.redbuf) {
            double flag = 0.0;
            HIP_TRY(h, hipMemcpy(&flag, src + k.n_pcof + 3, sizeof(double), hipMemcpyDeviceToHost));
            if (flag != 0.0) return fail(h, QGD_ERR_NUMERIC, "singular implicit step matrix L(t_n)");
        }
        if (grad && g) HIP_TRY(h, hipMemcpy(grad, g, sizeof(double) * k.n_pcof, hipMemcpyDeviceToHost));
        if (out3) HIP_TRY(h, hipMemcpy(out3, sc, 3 * sizeof(double), hipMemcpyDeviceToHost));
        return QGD_OK;
    }
    const size_t np = (size_t)k.n_pcof;
    if (!h->mirror_armed && !grad && src == k.redbuf && h->mirror_dev && h->mirror_ticket && !h->comm && h->chunks_eff == 1 && h->part_world == 1) {
        // an evaluation without a gradient (qgd_eval_forward): a one-workgroup kernel publishes the scalars the same way
        k.mirror_dev = h->mirror_dev; k.mirror_ticket = h->mirror_ticket; k.mirror_seq = ++h->mirror_seq;
        const int e = qgdk_mirror_scalars(&k);
        k.mirror_dev = nullptr;
        if (e) return fail(h, QGD_ERR_NO_DEVICE, "result mirror kernel failed to launch");
        h->mirror_armed = true;
    }
    if (h->mirror_armed && src == k.redbuf) {
        // the last kernel wrote the results into host memory itself: poll its sequence number (no copy packet, no wait for
        // the stream's completion signal -- the next evaluation's first launch overlaps the tail of this one's last kernel)
        h->mirror_armed = false;
        const volatile unsigned long long *seq = reinterpret_cast<const volatile unsigned long long *>(h->mirror_host + np + 5);
        const auto t0 = std::chrono::steady_clock::now();
        bool ok = true;
        for (unsigned spin = 1; *seq != h->mirror_seq; spin++) {
            if ((spin & 0xfffu) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 5.0) {
                // (never in a healthy run: fall back to the stream's own completion, which also surfaces a device error)
                HIP_TRY(h, hipStreamSynchronize(k.stream));
                ok = (*seq == h->mirror_seq);
                break;
            }
        }
        if (!ok) return fail(h, QGD_ERR_NO_DEVICE, "the evaluation finished without publishing its results (result mirror)");
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        const double *mh = h->mirror_host;
        if (mh[np + 4] != 0.0 || mh[np + 3] != 0.0) return fail(h, QGD_ERR_NUMERIC, "singular implicit step matrix L(t_n)");
        if (grad) memcpy(grad, mh, np * sizeof(double));
        if (out3) memcpy(out3, mh + np, 3 * sizeof(double));
        return QGD_OK;
    }
    HIP_TRY(h, hipMemcpyAsync(h->host_out, src, (np + 5) * sizeof(double), hipMemcpyDeviceToHost, k.stream));
    if (h->comm) { int rcw = comm_wait(h); if (rcw) return rcw; }      // collective evaluation: the wait is bounded
    else HIP_TRY(h, hipStreamSynchronize(k.stream));      // (spinning on hipStreamQuery instead: no difference, 359 us either way)
    int st = 0;
    if (src == k.redbuf) memcpy(&st, h->host_out + np + 4, sizeof(int));      // (a reduced buffer carries the flag as the double in front of it)
    // (host_out[np + 3]: the same flag as a double, summed over the ranks of a time-partitioned evaluation)
    if (st || h->host_out[np + 3] != 0.0) return fail(h, QGD_ERR_NUMERIC, "singular implicit step matrix L(t_n)");
    if (grad) memcpy(grad, h->host_out, np * sizeof(double));
    if (out3) memcpy(out3, h->host_out + np, 3 * sizeof(double));
    return QGD_OK;
}


// ---------------------------------------------------------------------------
// RCCL, bound at run time.  libqgd_hip.so has no link-time dependency on librccl: a single-GPU host never loads
// it.  dlopen("librccl.so.1") returns the copy already in the process when there is one (torch ships its own under
// the same SONAME), else the loader's search path, else /opt/rocm/lib.  QGD_RCCL_LIB names another file.
// ---------------------------------------------------------------------------
struct RcclApi {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;                          // (optional: older builds fall back to CommDestroy)
    ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t *) = nullptr;  // (optional)
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
    bool ok = false;
};

RcclApi load_rccl()
{
    RcclApi a;
    std::vector<std::string> names;
    if (const char *e = getenv("QGD_RCCL_LIB")) names.push_back(e);
    names.insert(names.end(), {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"});
    for (const auto &n : names) {
        a.lib = dlopen(n.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (a.lib) break;
        const char *de = dlerror();
        a.err += n + ": " + (de ? de : "?") + "; ";
    }
    if (!a.lib) { a.err = "RCCL could not be loaded (" + a.err + ")"; return a; }
    bool all = true;
    auto sym = [&](const char *name) { void *p = dlsym(a.lib, name); if (!p) { all = false; a.err += std::string(name) + " missing; "; } return p; };
    a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(sym("ncclGetUniqueId"));
    a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(sym("ncclCommInitRank"));
    a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(sym("ncclCommDestroy"));
    a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(sym("ncclAllReduce"));
    a.AllGather = reinterpret_cast<decltype(a.AllGather)>(sym("ncclAllGather"));
    a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(sym("ncclGetErrorString"));
    a.CommAbort = reinterpret_cast<decltype(a.CommAbort)>(dlsym(a.lib, "ncclCommAbort"));
    a.CommGetAsyncError = reinterpret_cast<decltype(a.CommGetAsyncError)>(dlsym(a.lib, "ncclCommGetAsyncError"));
    a.ok = all;
    return a;
}

RcclApi &rccl() { static RcclApi a = load_rccl(); return a; }

#define NCCL_TRY(h, expr)                                                                      \
    do {                                                                                       \
        ncclResult_t r__ = (expr);                                                             \
        if (r__ != ncclSuccess)                                                                \
            return fail((h), QGD_ERR_COMM, std::string(#expr) + ": " + rccl().GetErrorString(r__)); \
    } while (0)

// ---------------------------------------------------------------------------
// Failure mode of the collective calls.  A collective that one rank never enters blocks the others inside an RCCL
// kernel for good, so (1) the single host wait of a collective evaluation is a bounded hipStreamQuery loop that also
// polls ncclCommGetAsyncError, and (2) a rank that fails locally between two collectives, sees an asynchronous RCCL
// error or runs out of time ABORTS its communicator (ncclCommAbort makes the RCCL kernels on the stream return) and
// reports QGD_ERR_COMM.  The handle is left without a communicator: the host tears the job down (bench.py: the rank
// process exits non-zero) or builds a fresh communicator.  There is no retry inside the library.
// ---------------------------------------------------------------------------
void comm_abort(qgd_handle h)
{
    if (!h->comm) return;
    RcclApi &R = rccl();
    ncclComm_t c = h->comm;
    h->comm = nullptr; h->comm_rank = 0; h->comm_world = 1;
    if (R.CommAbort) (void)R.CommAbort(c); else (void)R.CommDestroy(c);
    (void)hipStreamSynchronize(h->k.stream);      // the library's own kernels behind the aborted collective drain normally
    (void)hipGetLastError();
}

int comm_failed(qgd_handle h, const std::string &why)
{
    comm_abort(h);
    return fail(h, QGD_ERR_COMM, why + "; the communicator of this handle was aborted (ncclCommAbort)");
}

// errors of a collective call that leave the OTHER ranks waiting in a collective this rank will not enter: device and
// launch failures, memory, RCCL itself.  Argument / state errors are raised before anything is launched (every rank
// gets them alike), and a singular step matrix travels with the reductions, so all ranks fail together without help.
int comm_local_error(qgd_handle h, int rc)
{
    if (rc == QGD_OK || !h->comm) return rc;
    if (rc == QGD_ERR_NO_DEVICE || rc == QGD_ERR_MEMORY || rc == QGD_ERR_COMM) {
        const std::string local = h->err;
        return comm_failed(h, "collective evaluation failed on rank " + std::to_string(h->comm_rank) + " (error " + std::to_string(rc) + ": " + local + ")");
    }
    return rc;
}

int comm_wait(qgd_handle h)
{
    RcclApi &R = rccl();
    const auto t0 = std::chrono::steady_clock::now();
    const int rank = h->comm_rank;
    for (unsigned spin = 1;; spin++) {
        const hipError_t q = hipStreamQuery(h->k.stream);
        if (q == hipSuccess) return QGD_OK;
        if (q != hipErrorNotReady) {
            (void)hipGetLastError();
            return comm_failed(h, std::string("rank ") + std::to_string(rank) + ": stream error while waiting for a collective evaluation: " + hipGetErrorString(q));
        }
        if (spin == 1 || (spin & 63u) == 0) {      // (the first look at the clock comes with the first unfinished query)
            if (R.CommGetAsyncError) {
                ncclResult_t ar = ncclSuccess;
                if (R.CommGetAsyncError(h->comm, &ar) == ncclSuccess && ar != ncclSuccess && ar != ncclInProgress)
                    return comm_failed(h, std::string("rank ") + std::to_string(rank) + ": RCCL reported an asynchronous error: " + R.GetErrorString(ar));
            }
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            if (ms > h->comm_timeout_ms)
                return comm_failed(h, "rank " + std::to_string(rank) + ": a collective evaluation did not complete within " + std::to_string((long long)h->comm_timeout_ms) +
                                      " ms (qgd_set_comm_timeout / QGD_COMM_TIMEOUT_MS): another rank failed or never made the call");
            if (ms > 2.0) sched_yield();      // (an evaluation takes well under a millisecond: past that, stop burning the core)
        }
    }
}

// which: the exchange buffers of qgd_exchange_buffer -- 0, 1 all-gather in place; 2 all-reduce(sum) of [grad | scalars];
// 3 all-reduce(sum) of the scalars {<w,R>, <w,T>, guard, flag} alone.  Issued on the handle's stream.
int comm_collective(qgd_handle h, int which)
{
    qgdk_ctx &k = h->k;
    RcclApi &R = rccl();
    if (h->comm_fail_at && h->comm_fail_at == which + 1) {      // (test hook: a local failure in front of this collective)
        h->comm_fail_at = 0;
        return fail(h, QGD_ERR_NO_DEVICE, "injected failure in front of collective " + std::to_string(which) + " (qgd_comm_debug_fail_at)");
    }
    const size_t pl = (size_t)k.Np * k.Np, hstep = (size_t)k.Np * 2 * k.cp;
    static const char *names[4] = {"comm_gather_fwd", "comm_gather_adj", "comm_reduce", "comm_reduce_scal"};
    PhaseTimer t(h, names[which]);
    if (which == 0) {
        const size_t chunk = 4 * pl;
        NCCL_TRY(h, R.AllGather(k.RX + (size_t)k.part_rank * chunk, k.RX, chunk, ncclDouble, h->comm, k.stream));
    } else if (which == 1) {
        const size_t chunk = 2 * hstep;
        NCCL_TRY(h, R.AllGather(k.phiRX + (size_t)k.part_rank * chunk, k.phiRX, chunk, ncclDouble, h->comm, k.stream));
    } else if (h->comm_shard == QGD_SHARD_TIME) {      // out of place: the rank's own sums stay where the kernels left them
        const size_t np = (size_t)k.n_pcof;
        if (which == 2) NCCL_TRY(h, R.AllReduce(k.redbuf, h->redglob, np + 4, ncclDouble, ncclSum, h->comm, k.stream));
        else NCCL_TRY(h, R.AllReduce(k.scal, h->redglob + np, 4, ncclDouble, ncclSum, h->comm, k.stream));
    } else {                                            // column blocks: the terminal condition reads the global overlaps in place
        if (which == 2) NCCL_TRY(h, R.AllReduce(k.redbuf, k.redbuf, (size_t)k.n_pcof + 4, ncclDouble, ncclSum, h->comm, k.stream));
        else NCCL_TRY(h, R.AllReduce(k.scal, k.scal, 4, ncclDouble, ncclSum, h->comm, k.stream));
    }
    return QGD_OK;
}

const double *comm_result(qgd_handle h) { return h->comm_shard == QGD_SHARD_TIME ? h->redglob : h->k.redbuf; }

bool same_pcof(qgd_handle h, const double *pcof, int n_pcof)
{
    if (h->history_stale) return false;     // (the last evaluation ran on the small-problem path: no stored history to reuse)
    return pcof ? ((size_t)n_pcof == h->fwd_pcof.size() && n_pcof > 0 && !memcmp(pcof, h->fwd_pcof.data(), sizeof(double) * n_pcof))
                : h->fwd_pcof.empty();
}

// forward sweep of a handle with a communicator: the rank's share + the exchange that completes it.
// Time windows: block products -> all-gather of the window products -> own history (+ overlaps on the last rank).
// Column blocks: the whole sweep on the own columns; the overlaps become global with the first reduction.
int comm_forward(qgd_handle h, const double *pcof, int n_pcof)
{
    int rc;
    if (h->comm_shard == QGD_SHARD_TIME) {
        if ((rc = forward_begin(h, pcof, n_pcof))) return rc;
        if ((rc = comm_collective(h, 0))) return rc;
        if ((rc = forward_end(h))) return rc;
    } else {
        if ((rc = forward_begin(h, pcof, n_pcof))) return rc;
        if ((rc = forward_end(h))) return rc;
    }
    if (pcof) h->fwd_pcof.assign(pcof, pcof + n_pcof); else h->fwd_pcof.clear();
    return QGD_OK;
}

// discrete_adjoint! of ONE SchrodingerProb spread over the ranks of the communicator (the reference's thread loop over
// columns, src/forward_evolution.jl:48,332; the global overlaps of src/infidelity.jl:13-17 and
// src/eval_grad_discrete_adjoint.jl:26-28 are what the collectives carry).  Every rank returns the full gradient and
// the global scalars.  The optional outputs cover what the rank owns: its window of time points (time shards,
// qgd_get_partition) or its columns (column shards).
int comm_discrete_adjoint_body(qgd_handle h, const double *pcof, int n_pcof, int history_precomputed, double *grad,
                               double *uv_history, double *lambda_history, double *adjoint_forcing, double *out3);

int comm_discrete_adjoint(qgd_handle h, const double *pcof, int n_pcof, int history_precomputed, double *grad,
                          double *uv_history, double *lambda_history, double *adjoint_forcing, double *out3)
{
    // what every rank gets alike is refused before anything is launched (no collective is left half-entered)
    if (pcof && n_pcof != h->k.n_pcof) return fail(h, QGD_ERR_ARGUMENT, "length of pcof does not match the control basis");
    if (!pcof && !h->have_tables && h->k.n_ops > 0) return fail(h, QGD_ERR_STATE, "no control tables: call qgd_set_control_tables or pass pcof");
    if (history_precomputed && !h->forward_valid) return fail(h, QGD_ERR_STATE, "history_precomputed without a previous forward evaluation");
    return comm_local_error(h, comm_discrete_adjoint_body(h, pcof, n_pcof, history_precomputed, grad, uv_history, lambda_history, adjoint_forcing, out3));
}

int comm_discrete_adjoint_body(qgd_handle h, const double *pcof, int n_pcof, int history_precomputed, double *grad,
                               double *uv_history, double *lambda_history, double *adjoint_forcing, double *out3)
{
    qgdk_ctx &k = h->k;
    int rc;
    struct CopyGuard { qgd_handle h; ~CopyGuard() { (void)finish_copies(h); h->defer_terminal = false; h->lambda_out = nullptr; } } guard{h};
    if (history_precomputed && !h->forward_valid)
        return fail(h, QGD_ERR_STATE, "history_precomputed without a previous forward evaluation");
    const bool reuse = history_precomputed && same_pcof(h, pcof, n_pcof);
    const bool time = h->comm_shard == QGD_SHARD_TIME;
    if (reuse) {
        // time shards: the rank's own guard sum is still in scal (the reductions are out of place); column shards: the
        // scalars on the device are the global ones of the call that made the history
        if (time && k.part_rank == k.part_world - 1) { PhaseTimer t(h, "terminal"); K_TRY(h, qgdk_terminal(&k, 1)); }
    } else {
        // (time shards: on the rank that owns the final time the overlaps and y_N ride in the first adjoint launch, as in
        //  the single-GPU evaluation, instead of a k_terminal launch of their own)
        h->defer_terminal = time && k.have_target && k.part_rank == k.part_world - 1 && qgdk_terminal_can_fuse(&k) != 0;
        if ((rc = comm_forward(h, pcof, n_pcof))) return rc;
        if (!time && (rc = comm_collective(h, 3))) return rc;      // <w_N,R>, <w_N,T>, guard: global before the terminal condition
    }
    if (!time) { PhaseTimer t(h, "terminal"); K_TRY(h, qgdk_terminal_given(&k)); }
    if (adjoint_forcing && (rc = copy_panels_out(h, k.forcing, &h->stage_f, adjoint_forcing, 1, 0))) return rc;
    if (uv_history) {
        if (!h->derivs_valid) { PhaseTimer t(h, "derivs"); K_TRY(h, qgdk_derivs(&k)); h->derivs_valid = true; }
        if ((rc = copy_history_out(h, uv_history))) return rc;
    }
    if ((rc = adjoint_begin(h))) return rc;
    if (time && (rc = comm_collective(h, 1))) return rc;
    h->lambda_out = lambda_history;
    rc = adjoint_end(h);
    h->lambda_out = nullptr;
    if (rc) return rc;
    // column shards: the scalars are global on every rank already -- all ranks but the first contribute zeros
    if (!time && h->comm_rank != 0) HIP_TRY(h, hipMemsetAsync(k.scal, 0, 3 * sizeof(double), k.stream));
    if ((rc = comm_collective(h, 2))) return rc;
    if ((rc = fetch_results(h, grad, out3, comm_result(h)))) return rc;
    return finish_copies(h);
}

int comm_eval_forward_body(qgd_handle h, const double *pcof, int n_pcof, double *uv_history, double *out3);

int comm_eval_forward(qgd_handle h, const double *pcof, int n_pcof, double *uv_history, double *out3)
{
    if (pcof && n_pcof != h->k.n_pcof) return fail(h, QGD_ERR_ARGUMENT, "length of pcof does not match the control basis");
    if (pcof && !h->have_basis) return fail(h, QGD_ERR_STATE, "qgd_set_control_basis must be called before passing pcof");
    if (!pcof && !h->have_tables && h->k.n_ops > 0) return fail(h, QGD_ERR_STATE, "no control tables: call qgd_set_control_tables or pass pcof");
    return comm_local_error(h, comm_eval_forward_body(h, pcof, n_pcof, uv_history, out3));
}

int comm_eval_forward_body(qgd_handle h, const double *pcof, int n_pcof, double *uv_history, double *out3)
{
    qgdk_ctx &k = h->k;
    int rc = comm_forward(h, pcof, n_pcof);
    if (rc) return rc;
    if (uv_history) {
        { PhaseTimer t(h, "derivs"); K_TRY(h, qgdk_derivs(&k)); }
        h->derivs_valid = true;
        if ((rc = copy_history_out(h, uv_history, h->save_every))) return rc;
    }
    K_TRY(h, qgdk_flag_to_scal(&k));      // a singular step matrix on ANY rank fails the call on every rank
    if ((rc = comm_collective(h, 3))) { (void)finish_copies(h); return rc; }
    if ((rc = fetch_results(h, nullptr, out3, comm_result(h)))) { (void)finish_copies(h); return rc; }
    return finish_copies(h);
}

}  // namespace

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
    // operators of multi_qudit_systems.jl have 1 + 2*subsystems); QGD_DENSE_OPS=1 keeps the MFMA path.
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
        k.use_sparse = h->sparse_available && !getenv("QGD_DENSE_OPS");
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
                return !offs.empty() && (int)offs.size() <= zmax && !getenv("QGD_ELL_ROW_PACKED");
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
    CREATE_RC(dev_alloc(h, h->static_bufs, &k.status, (size_t)2));
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
    if (h->k.stream) (void)hipStreamSynchronize(h->k.stream);
    drop_graph(h);
    if (h->comm) (void)qgd_comm_destroy(h);
    if (h->copy_stream) { (void)hipStreamSynchronize(h->copy_stream); (void)hipStreamDestroy(h->copy_stream); }
    if (h->copy_stream2) { (void)hipStreamSynchronize(h->copy_stream2); (void)hipStreamDestroy(h->copy_stream2); }
    for (int i = 0; i < qgd_handle_s::MAX_CHUNKS; i++) {
        if (h->pipe_stream[i]) { (void)hipStreamSynchronize(h->pipe_stream[i]); (void)hipStreamDestroy(h->pipe_stream[i]); }
        if (h->pipe_built[i]) (void)hipEventDestroy(h->pipe_built[i]);
        if (h->pipe_done[i]) (void)hipEventDestroy(h->pipe_done[i]);
    }
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
    h->have_basis = false; h->forward_valid = false; h->derivs_valid = false; h->fwd_pcof.clear();
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
        h->have_tables = true; h->forward_valid = false;
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

int qgd_eval_forward(qgd_handle h, const double *pcof, int32_t n_pcof, double *uv_history, double *out3)
{
    if (!h) return QGD_ERR_ARGUMENT;
    HIP_TRY(h, hipSetDevice(h->device));
    NEED_GRID(h);
    qgdk_ctx &k = h->k;
    if (h->comm) return comm_eval_forward(h, pcof, n_pcof, uv_history, out3);
    if (h->chunks_eff > 1 && uv_history) {      // the history of a chunked grid comes out window by window
        int rcw = chunked_forward(h, pcof, n_pcof, uv_history, h->save_every);
        return rcw ? rcw : fetch_results(h, nullptr, out3);
    }
    if (!uv_history && tiny_applies(h, pcof, n_pcof)) return tiny_evaluate(h, pcof, n_pcof, false, nullptr, out3);
    int rc = run_forward(h, pcof, n_pcof);
    if (rc) return rc;
    if (uv_history) {
        { PhaseTimer t(h, "derivs"); K_TRY(h, qgdk_derivs(&k)); }
        h->derivs_valid = true;
    }
    if (uv_history && (rc = copy_history_out(h, uv_history, h->save_every))) return rc;
    if ((rc = fetch_results(h, nullptr, out3))) { (void)finish_copies(h); return rc; }
    return finish_copies(h);
}

int qgd_discrete_adjoint(qgd_handle h, const double *pcof, int32_t n_pcof, int32_t history_precomputed,
                         double *grad, double *uv_history, double *lambda_history, double *adjoint_forcing,
                         double *out3)
{
    if (!h || !grad) return fail(h, QGD_ERR_ARGUMENT, "null argument");
    HIP_TRY(h, hipSetDevice(h->device));
    NEED_GRID(h);
    qgdk_ctx &k = h->k;
    if (!k.have_target) return fail(h, QGD_ERR_STATE, "qgd_set_target must be called before qgd_discrete_adjoint");
    if (!h->have_basis) return fail(h, QGD_ERR_STATE, "qgd_set_control_basis must be called before qgd_discrete_adjoint");
    if (h->comm) return comm_discrete_adjoint(h, pcof, n_pcof, history_precomputed, grad, uv_history, lambda_history, adjoint_forcing, out3);
    if (h->part_world != 1) return fail(h, QGD_ERR_STATE, "partitioned handle: use the qgd_dist_* entry points, or give the handle a communicator (qgd_comm_init_rccl)");
    int rc;
    if (h->chunks_eff > 1) {      // bounded-memory time grid: forward pass over the windows, adjoint pass back over them
        if (history_precomputed && !h->forward_valid) return fail(h, QGD_ERR_STATE, "history_precomputed without a previous forward evaluation");
        // (uv_history is an output: a reused forward pass would have nothing to copy it from, so the pass is redone)
        if ((uv_history || !(history_precomputed && same_pcof(h, pcof, n_pcof))) && (rc = chunked_forward(h, pcof, n_pcof, uv_history))) return rc;
        if ((rc = chunked_adjoint(h, lambda_history, adjoint_forcing))) return rc;
        return fetch_results(h, grad, out3);
    }
    if (!uv_history && !lambda_history && !adjoint_forcing && tiny_applies(h, pcof, n_pcof)) {
        if (history_precomputed && !h->forward_valid) return fail(h, QGD_ERR_STATE, "history_precomputed without a previous forward evaluation");
        return tiny_evaluate(h, pcof, n_pcof, true, grad, out3);      // (four launches redo the sweep faster than the stored one could be reused)
    }
    // full evaluation, nothing but [grad | scalars] coming back, no event bracketing: replay the captured launch sequence
    const bool graph_ok = !h->graph_off && !history_precomputed && !uv_history && !lambda_history && !adjoint_forcing &&
                          !h->timing && k.redbuf && h->host_out && h->host_in && h->host_out_len >= (size_t)n_pcof && pcof;
    if (graph_ok) {
        if (n_pcof != k.n_pcof) return fail(h, QGD_ERR_ARGUMENT, "length of pcof does not match the control basis");
        if (!h->graph_exec && ++h->graph_calls >= 3) {      // the first calls set function attributes and fill caches
            hipGraph_t g = nullptr;
            bool ok = hipStreamBeginCapture(k.stream, hipStreamCaptureModeRelaxed) == hipSuccess;
            if (ok) {
                h->defer_terminal = qgdk_terminal_can_fuse(&k) != 0;
                rc = run_forward(h, pcof, n_pcof);
                if (!rc) rc = adjoint_begin(h);
                if (!rc) rc = adjoint_end(h);
                if (!rc && hipMemcpyAsync(h->host_out, k.redbuf, ((size_t)k.n_pcof + 5) * sizeof(double), hipMemcpyDeviceToHost, k.stream) != hipSuccess) rc = 1;
                ok = (hipStreamEndCapture(k.stream, &g) == hipSuccess) && !rc && g;
                if (ok) ok = hipGraphInstantiate(&h->graph_exec, g, nullptr, nullptr, 0) == hipSuccess;
                if (ok) h->graph = g; else { if (g) (void)hipGraphDestroy(g); h->graph_exec = nullptr; }
            }
            (void)hipGetLastError();
            if (!ok) {      // e.g. the legacy default stream cannot be captured: plain launches from now on
                h->graph_off = true;
                h->forcing_zero = false;      // (the capture only RECORDED the guard clear: the plain path must run it)
            }
        }
        if (h->graph_exec) {
            memcpy(h->host_in, pcof, sizeof(double) * n_pcof);
            HIP_TRY(h, hipGraphLaunch(h->graph_exec, k.stream));
            HIP_TRY(h, hipStreamSynchronize(k.stream));
            h->forward_valid = true;
            h->fwd_pcof.assign(pcof, pcof + n_pcof);
            h->derivs_valid = qgdk_gradient_needs_derivs(&k) != 0;
            const size_t np = (size_t)k.n_pcof;
            int st;
            memcpy(&st, h->host_out + np + 4, sizeof(int));
            if (st) return fail(h, QGD_ERR_NUMERIC, "singular implicit step matrix L(t_n)");
            memcpy(grad, h->host_out, np * sizeof(double));
            if (out3) memcpy(out3, h->host_out + np, 3 * sizeof(double));
            return QGD_OK;
        }
    }
    // QGD_HOST_TRACE=1: host-side time stamps of the stages of this call (us since entry) on stderr
    static const bool host_trace = getenv("QGD_HOST_TRACE") != nullptr;
    const auto t_entry = std::chrono::steady_clock::now();
    auto stamp = [&](const char *what) {
        if (host_trace) fprintf(stderr, "[qgd host] %-22s %8.1f us\n", what,
                                std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_entry).count());
    };
    struct CopyGuard { qgd_handle h; ~CopyGuard() { (void)finish_copies(h); h->defer_terminal = false; } } guard{h};   // no copy (and no deferred terminal condition) outlives the call
    // history_precomputed: the reference differentiates the history it is GIVEN with the pcof it is given
    // (eval_grad_discrete_adjoint.jl:118-124).  The device keeps its own copy of the last forward sweep; it is
    // reused only when it was computed from this very pcof, otherwise the sweep is simply redone.
    if (history_precomputed && !h->forward_valid)
        return fail(h, QGD_ERR_STATE, "history_precomputed without a previous forward evaluation");
    const bool reuse = history_precomputed && same_pcof(h, pcof, n_pcof);
    if (reuse) {
        // the terminal right-hand side may not have been written if the target was set later
        { PhaseTimer t(h, "terminal"); K_TRY(h, qgdk_terminal(&k, 1)); }
    } else {
        h->defer_terminal = k.have_target && qgdk_terminal_can_fuse(&k) != 0;
        rc = run_forward(h, pcof, n_pcof);
        if (rc) { h->defer_terminal = false; return rc; }
    }
    stamp("forward launched");
    // The downloads run on the copy stream beside the adjoint sweep and are what bounds this form of the call (PCIe):
    // the guard forcing goes first -- it is final once the forward sweep is (eval_grad_discrete_adjoint.jl:732-752) and
    // keeps the link busy while the stage derivatives of the state history are still being computed and laid out
    if (adjoint_forcing && (rc = copy_panels_out(h, k.forcing, &h->stage_f, adjoint_forcing, 1, 0))) return rc;
    stamp("forcing copy issued");
    if (uv_history) {
        if (!h->derivs_valid) { PhaseTimer t(h, "derivs"); K_TRY(h, qgdk_derivs(&k)); h->derivs_valid = true; }
        if ((rc = copy_history_out(h, uv_history))) return rc;
    }
    stamp("history copy issued");
    if ((rc = adjoint_begin(h))) return rc;
    h->lambda_out = lambda_history;
    // (result mirror: single GPU, resident grid; with event bracketing on too -- qgd_get_timings synchronises the stream itself)
    const bool mirror = h->mirror_dev && h->mirror_ticket && k.redbuf && k.n_ops > 0;
    if (mirror) { k.mirror_dev = h->mirror_dev; k.mirror_ticket = h->mirror_ticket; k.mirror_seq = ++h->mirror_seq; }
    rc = adjoint_end(h);
    k.mirror_dev = nullptr;
    h->lambda_out = nullptr;
    if (rc) return rc;
    h->mirror_armed = mirror;
    stamp("adjoint launched");
    if ((rc = fetch_results(h, grad, out3))) return rc;
    stamp("results fetched");
    rc = finish_copies(h);
    stamp("copies finished");
    return rc;
}

int qgd_eval_forward_forced(qgd_handle h, const double *pcof, int32_t n_pcof, const double *forcing,
                            double *uv_history, double *out3)
{
    if (h) drop_graph(h);
    if (!h || !forcing) return fail(h, QGD_ERR_ARGUMENT, "null argument");
    HIP_TRY(h, hipSetDevice(h->device));
    NEED_GRID(h);
    qgdk_ctx &k = h->k;
    if (h->part_world != 1) return fail(h, QGD_ERR_STATE, "partitioned handle: the forced forward sweep is single-GPU");
    if (h->chunks_eff > 1) return chunked_forward_forced(h, pcof, n_pcof, forcing, uv_history, out3);
    int rc = forward_begin(h, pcof, n_pcof);
    if (rc) return rc;
    if ((rc = forcing_buffers(h, (size_t)k.nt, (size_t)k.scan_blocks))) return rc;
    if ((rc = upload_forcing(h, forcing, (size_t)k.nt, 0))) return rc;
    { PhaseTimer t(h, "forcing_terms"); K_TRY(h, qgdk_forcing_terms(&k)); }
    { PhaseTimer t(h, "sweep_forced"); K_TRY(h, qgdk_forcing_sweep(&k)); }
    // (the stand-alone guard kernel stores ONE partial penalty per time point; forward_begin sized the fixed-order sum for
    //  the history pass that fuses the guard work -- fewer, per-block partials -- which this sweep does not run: round 3's
    //  sum added only the first of them and returned a guard penalty that was too small)
    k.gpart_n = k.nt;
    { PhaseTimer t(h, "guard"); K_TRY(h, qgdk_guard_kernel(&k)); }
    { PhaseTimer t(h, "terminal"); K_TRY(h, qgdk_terminal(&k, k.have_target)); }
    h->forward_valid = false;      // this history is not the one the adjoint sweep differentiates
    if (uv_history) {
        { PhaseTimer t(h, "derivs"); K_TRY(h, qgdk_derivs(&k)); }
        K_TRY(h, qgdk_forcing_add_derivs(&k));     // w_j = D_j w_0 + E_j
        h->derivs_valid = false;
    }
    if (uv_history && (rc = copy_history_out(h, uv_history, h->save_every))) return rc;
    if ((rc = fetch_results(h, nullptr, out3))) { (void)finish_copies(h); return rc; }
    return finish_copies(h);
}

// buffers of the forced gradient for (up to) nt time points and B scan blocks
static int forced_buffers(qgd_handle h, size_t nt, size_t B)
{
    qgdk_ctx &k = h->k;
    const size_t hstep = (size_t)k.Np * 2 * k.cp, NB = (size_t)k.n_ops * 2 * k.m;
    const size_t cpS = (size_t)k.n_pcof * k.cp, hstepS = (size_t)k.Np * 2 * cpS;
    const size_t key = (nt * 1000003u + (size_t)k.n_pcof) * 4099u + B;
    int rc;
    if (h->forced_key != key) {
        free_pool(h->forced_bufs); h->forced_key = 0;
        if ((rc = dev_alloc(h, h->forced_bufs, &k.fs_BR, nt * NB * hstep))) return rc;
        if ((rc = dev_alloc(h, h->forced_bufs, &k.fs_BL, nt * NB * hstep))) return rc;
        if ((rc = dev_alloc(h, h->forced_bufs, &k.fs_phi, B * hstepS))) return rc;
        if ((rc = dev_alloc(h, h->forced_bufs, &k.fs_bnd, (B + 1) * hstepS))) return rc;
        if ((rc = dev_alloc(h, h->forced_bufs, &k.fs_gacc, (size_t)k.n_pcof + 1))) return rc;
        h->fsc_forced = nullptr;      // (the 2m+2 work panels of k_forced_basis: LDS up to 150 KB, else an HBM slab per workgroup)
        if (qgdk_forced_lds(k.Np, k.m) > 150 * 1024 &&
            (rc = dev_alloc(h, h->forced_bufs, &h->fsc_forced, nt * (size_t)(k.cp / 8) * (size_t)(2 * k.m + 2) * k.Np * 16))) return rc;
        h->forced_key = key;
    }
    k.fs_scratch = h->fsc_forced;
    return QGD_OK;
}

int qgd_eval_grad_forced(qgd_handle h, const double *pcof, int32_t n_pcof, double *grad)
{
    if (h) drop_graph(h);
    if (!h || !grad) return fail(h, QGD_ERR_ARGUMENT, "null argument");     // (pcof may be NULL when the tables were set directly)
    HIP_TRY(h, hipSetDevice(h->device));
    NEED_GRID(h);
    qgdk_ctx &k = h->k;
    if (!k.have_target) return fail(h, QGD_ERR_STATE, "qgd_set_target must be called before qgd_eval_grad_forced");
    if (!h->have_basis) return fail(h, QGD_ERR_STATE, "qgd_set_control_basis must be called before qgd_eval_grad_forced");
    if (h->part_world != 1) return fail(h, QGD_ERR_STATE, "partitioned handle: the forced gradient is single-GPU");
    int rc;
    if ((rc = run_forward(h, pcof, n_pcof))) return rc;      // (a windowed grid: every window, the state at each window start kept)
    const size_t hstep = (size_t)k.Np * 2 * k.cp;
    const size_t cpS = (size_t)k.n_pcof * k.cp, hstepS = (size_t)k.Np * 2 * cpS;
    size_t nt = k.nt, B = k.scan_blocks;
    if (h->chunks_eff == 1) {
        if (!h->derivs_valid) { PhaseTimer t(h, "derivs"); K_TRY(h, qgdk_derivs(&k)); h->derivs_valid = true; }
        if ((rc = forced_buffers(h, nt, B))) return rc;
        HIP_TRY(h, hipMemsetAsync(k.fs_bnd, 0, hstepS * sizeof(double), k.stream));
        HIP_TRY(h, hipMemsetAsync(k.fs_gacc, 0, ((size_t)k.n_pcof + 1) * sizeof(double), k.stream));
        { PhaseTimer t(h, "forced_basis"); K_TRY(h, qgdk_forced_basis(&k)); }
        { PhaseTimer t(h, "forced_sweeps"); K_TRY(h, qgdk_forced_chains(&k)); }
    } else {
        // Windows in order: each forms its matrices and forward history again from its stored start state (as the adjoint pass
        // does), the sensitivities of all parameters continue from where the previous window left them, the guard part of the
        // gradient accumulates.  (eval_grad_forced.jl:17-194 keeps no matrices either: one forced sweep per parameter.)
        const std::vector<double> pc(h->fwd_pcof);
        const double *pp = pc.empty() ? nullptr : pc.data();
        size_t nt0 = 0, B0 = 0;
        for (int r = 0; r < h->chunks_eff; r++) {
            if ((rc = chunk_forward(h, pp, (int)pc.size(), r, true))) return rc;
            if (r == 0) {
                nt0 = (size_t)k.nt; B0 = (size_t)k.scan_blocks;      // (the first window is the longest)
                if ((rc = forced_buffers(h, nt0, B0))) return rc;
                HIP_TRY(h, hipMemsetAsync(k.fs_bnd, 0, hstepS * sizeof(double), k.stream));
                HIP_TRY(h, hipMemsetAsync(k.fs_gacc, 0, ((size_t)k.n_pcof + 1) * sizeof(double), k.stream));
            }
            k.fs_scratch = h->fsc_forced;
            { PhaseTimer t(h, "derivs"); K_TRY(h, qgdk_derivs(&k)); }
            { PhaseTimer t(h, "forced_basis"); K_TRY(h, qgdk_forced_basis(&k)); }
            { PhaseTimer t(h, "forced_sweeps"); K_TRY(h, qgdk_forced_chains(&k)); }
            nt = k.nt; B = k.scan_blocks;
            if (r + 1 < h->chunks_eff)      // s at the start of the next window
                HIP_TRY(h, hipMemcpyAsync(k.fs_bnd, k.fs_bnd + B * hstepS, hstepS * sizeof(double), hipMemcpyDeviceToDevice, k.stream));
        }
        h->derivs_valid = false;
    }
    if ((rc = check_status(h))) return rc;
    std::vector<double> sN(hstepS), gacc(k.n_pcof), scal(4), wN;
    HIP_TRY(h, hipMemcpy(sN.data(), k.fs_bnd + B * hstepS, hstepS * sizeof(double), hipMemcpyDeviceToHost));
    if (k.cost_type) {     // :Tracking / :Norm need the final state itself (eval_grad_forced.jl:160-163)
        wN.resize(hstep);
        HIP_TRY(h, hipMemcpy(wN.data(), k.hist + (nt - 1) * hstep, hstep * sizeof(double), hipMemcpyDeviceToHost));
    }
    HIP_TRY(h, hipMemcpy(gacc.data(), k.fs_gacc, sizeof(double) * k.n_pcof, hipMemcpyDeviceToHost));
    HIP_TRY(h, hipMemcpy(scal.data(), k.scal, 3 * sizeof(double), hipMemcpyDeviceToHost));
    // d(infidelity) = -(2/N_ess^2) (<w_N,R> <s_N,R> + <w_N,T> <s_N,T>), T = [R_im; -R_re] (infidelity.jl:13-17)
    const size_t N = k.N, PWs = 2 * cpS;
    const double a = scal[0], b = scal[1], f = -2.0 / ((double)k.n_ess * k.n_ess);
    for (int p = 0; p < k.n_pcof; p++) {
        double sR = 0.0, sT = 0.0, sW = 0.0;
        for (int col = 0; col < k.c; col++)
            for (size_t i = 0; i < N; i++) {
                const size_t o = panel_index((int)i, p * k.cp + col, (int)PWs);
                const double sre = sN[o], sim = sN[o + 8];
                const double rre = h->target_host[i + 2 * N * col], rim = h->target_host[N + i + 2 * N * col];
                sR += sre * rre + sim * rim;
                sT += sre * rim - sim * rre;
                if (k.cost_type) {      // d(0.5 |w_N - R|^2) = <s_N, w_N - R>,  d(0.5 |w_N|^2) = <s_N, w_N>
                    const size_t ow = panel_index((int)i, col, 2 * k.cp);
                    const double dre = wN[ow] - (k.cost_type == QGD_COST_TRACKING ? rre : 0.0);
                    const double dim = wN[ow + 8] - (k.cost_type == QGD_COST_TRACKING ? rim : 0.0);
                    sW += sre * dre + sim * dim;
                }
            }
        grad[p] = (k.cost_type ? sW : f * (a * sR + b * sT)) + gacc[p];
    }
    return QGD_OK;
}

int qgd_eval_adjoint(qgd_handle h, const double *pcof, int32_t n_pcof, const double *terminal_condition,
                     const double *forcing, double *lambda_history)
{
    if (h) drop_graph(h);
    if (!h || !terminal_condition || !lambda_history) return fail(h, QGD_ERR_ARGUMENT, "null argument");
    HIP_TRY(h, hipSetDevice(h->device));
    NEED_GRID(h);
    qgdk_ctx &k = h->k;
    if (h->part_world != 1) return fail(h, QGD_ERR_STATE, "partitioned handle: eval_adjoint is single-GPU");
    if (h->chunks_eff > 1) return chunked_eval_adjoint(h, pcof, n_pcof, terminal_condition, forcing, lambda_history);
    int rc = forward_begin(h, pcof, n_pcof);          // tables, L/R, inverses, propagators, block propagators
    if (rc) return rc;
    // the second scan level (super-block propagators) is produced by the forward boundary phase
    { PhaseTimer t(h, "sweep_forward2"); K_TRY(h, qgdk_forward_finish(&k)); }
    const size_t Np = k.Np, PWc = 2 * k.cp, hstep = Np * PWc, nt = k.nt, N = k.N, n2 = 2 * N, m = k.m;
    // forcing [2N, nt, c] and terminal condition [2N, c] into panel layout
    std::vector<double> f(nt * hstep, 0.0), lamN(hstep, 0.0);
    for (size_t col = 0; col < (size_t)k.c; col++) {
        for (size_t i = 0; i < N; i++) {
            size_t o = panel_index((int)i, (int)col, (int)PWc);
            lamN[o] = terminal_condition[i + n2 * col];
            lamN[o + 8] = terminal_condition[N + i + n2 * col];
        }
        if (forcing)
            for (size_t n = 0; n < nt; n++) for (size_t i = 0; i < N; i++) {
                size_t o = n * hstep + panel_index((int)i, (int)col, (int)PWc);
                const double *src = forcing + (col * nt + n) * n2;
                f[o] = src[i]; f[o + 8] = src[N + i];
            }
    }
    HIP_TRY(h, hipMemcpyAsync(k.forcing, f.data(), f.size() * sizeof(double), hipMemcpyHostToDevice, k.stream));
    h->forcing_zero = false;
    HIP_TRY(h, hipMemcpyAsync(k.lam + (nt - 1) * hstep, lamN.data(), hstep * sizeof(double), hipMemcpyHostToDevice, k.stream));
    // y_N = L(t_N)^H lambda_N  (the terminal condition is lambda itself here, forward_evolution.jl:411-414)
    K_TRY(h, qgdk_apply_LH(&k));
    if ((rc = adjoint_begin(h))) return rc;
    { PhaseTimer t(h, "sweep_adjoint2"); K_TRY(h, qgdk_adjoint_finish(&k)); }
    { PhaseTimer t(h, "lambda"); K_TRY(h, qgdk_lambda(&k)); }
    if ((rc = check_status(h))) return rc;
    if (h->lambda_derivs) {      // the reference's derivative columns too (forward_evolution.jl:427-433, :471-480)
        HIP_TRY(h, hipMemcpyAsync(k.lam + (nt - 1) * hstep, lamN.data(), hstep * sizeof(double), hipMemcpyHostToDevice, k.stream));
        rc = copy_lambda_full_out(h, lambda_history);
        const int rc2 = finish_copies(h);
        h->forward_valid = false;
        return rc ? rc : rc2;
    }
    std::vector<double> l(nt * hstep);
    HIP_TRY(h, hipMemcpy(l.data(), k.lam, l.size() * sizeof(double), hipMemcpyDeviceToHost));
    memset(lambda_history, 0, sizeof(double) * n2 * (m + 1) * nt * k.c);
    for (size_t col = 0; col < (size_t)k.c; col++) for (size_t n = 1; n < nt; n++) {
        double *dst = lambda_history + ((col * nt + n) * (m + 1)) * n2;
        const double *src = (n == nt - 1) ? lamN.data() : l.data() + n * hstep;   // lambda_N is the given one
        for (size_t i = 0; i < N; i++) {
            size_t o = panel_index((int)i, (int)col, (int)PWc);
            dst[i] = src[o]; dst[N + i] = src[o + 8];
        }
    }
    h->forward_valid = false;   // the state history was not computed
    return QGD_OK;
}

int qgd_apply_hamiltonian(qgd_handle h, int32_t time_index, int32_t deriv_order, int32_t use_adjoint,
                          const double *in, double *out)
{
    if (h) drop_graph(h);
    if (!h || !in || !out) return fail(h, QGD_ERR_ARGUMENT, "null argument");
    HIP_TRY(h, hipSetDevice(h->device));
    NEED_GRID(h);
    qgdk_ctx &k = h->k;
    NEEDS_RESIDENT_GRID(h, "qgd_apply_hamiltonian");
    if (time_index < 0 || time_index >= k.nt || deriv_order < 0 || deriv_order > k.m)
        return fail(h, QGD_ERR_ARGUMENT, "time index or derivative order out of range");
    const size_t PWc = 2 * k.cp, cnt = (size_t)k.Np * PWc;
    std::vector<double> p(cnt, 0.0), q(cnt, 0.0);
    for (int col = 0; col < k.c; col++) for (int i = 0; i < k.N; i++) {
        size_t o = panel_index(i, col, (int)PWc);
        p[o] = in[i + (size_t)2 * k.N * col]; p[o + 8] = in[k.N + i + (size_t)2 * k.N * col];
    }
    double *din = nullptr, *dout = nullptr;
    HIP_TRY(h, hipMalloc((void **)&din, cnt * sizeof(double)));
    HIP_TRY(h, hipMalloc((void **)&dout, cnt * sizeof(double)));
    HIP_TRY(h, hipMemcpy(din, p.data(), cnt * sizeof(double), hipMemcpyHostToDevice));
    int kr = qgdk_apply(&k, din, dout, time_index, deriv_order, use_adjoint ? -1.0 : 1.0);
    hipError_t e = hipStreamSynchronize(k.stream);
    if (!kr && e == hipSuccess) e = hipMemcpy(q.data(), dout, cnt * sizeof(double), hipMemcpyDeviceToHost);
    (void)hipFree(din); (void)hipFree(dout);
    if (kr || e != hipSuccess) return fail(h, QGD_ERR_NO_DEVICE, "apply kernel failed");
    for (int col = 0; col < k.c; col++) for (int i = 0; i < k.N; i++) {
        size_t o = panel_index(i, col, (int)PWc);
        out[i + (size_t)2 * k.N * col] = q[o]; out[k.N + i + (size_t)2 * k.N * col] = q[o + 8];
    }
    return QGD_OK;
}

int qgd_get_intermediate(qgd_handle h, const char *name, double *out, size_t capacity, size_t *needed)
{
    if (!h || !name) return fail(h, QGD_ERR_ARGUMENT, "null argument");
    HIP_TRY(h, hipSetDevice(h->device));
    NEED_GRID(h);
    qgdk_ctx &k = h->k;
    const size_t Np = k.Np, N = k.N, nt = k.nt, PW = 2 * Np, panel = Np * PW, pl = Np * Np;
    std::string s(name);
    size_t need = 0;
    if (s == "L" || s == "R" || s == "Linv" || s == "P") need = nt * N * N * 2;
    else if (s == "sigma") need = nt * (size_t)k.n_ops * k.m * 2;
    else if (s == "tables") need = nt * (size_t)(k.m + 1) * k.n_ops * 2;
    else if (s == "repivoted") need = 1;
    else if (s == "selection") need = 4;
    else if (s == "small_path") need = 1;
    else return fail(h, QGD_ERR_ARGUMENT, "unknown intermediate '" + s + "'");
    if (needed) *needed = need;
    if (!out) return QGD_OK;
    if (capacity < need) return fail(h, QGD_ERR_ARGUMENT, "buffer too small");
    if (s == "small_path") { out[0] = h->history_stale ? 1.0 : 0.0; return QGD_OK; }      // did the LAST evaluation run on the small-problem path
    if (s == "selection") {      // which kernel families this problem runs on (tests assert that a shape selects what it is meant to)
        out[0] = k.use_sparse ? 2.0 : (k.dense_gemm ? 1.0 : 0.0);      // 2 sparse (ELL), 1 N > 64 GEMM-style kernels, 0 dense N <= 64
        out[1] = (k.dense_gemm && !k.use_sparse) ? (double)qgdk_dense_sigma_form(&k) : -1.0;
        out[2] = (k.dense_gemm && k.binv) ? 1.0 : 0.0;               // block Gauss-Jordan inverse over 64-column blocks
        out[3] = (double)h->chunks_eff;
        return QGD_OK;
    }
    NEEDS_RESIDENT_GRID(h, "qgd_get_intermediate");
    if (h->history_stale && s != "repivoted" && !h->tiny_pcof.empty()) {
        // the last evaluation ran on the small-problem path, which keeps no intermediates: the same evaluation once more on
        // the general path (diagnostics only)
        const std::vector<double> pc = h->tiny_pcof;
        int rcs = run_forward(h, pc.data(), (int)pc.size());
        if (!rcs && h->tiny_was_gradient && k.have_target) { rcs = adjoint_begin(h); if (!rcs) rcs = adjoint_end(h); }
        if (rcs) return rcs;
    }
    HIP_TRY(h, hipStreamSynchronize(k.stream));
    if (s == "repivoted") {     // workgroups of the last inverse launch whose static-pivot attempt was redone with partial pivoting
        int v[2] = {0, 0};
        HIP_TRY(h, hipMemcpy(v, k.status, 2 * sizeof(int), hipMemcpyDeviceToHost));
        out[0] = (double)v[1];
        return QGD_OK;
    }
    if (s == "sigma") {      // (the column groups' planes of the N <= 64 gradient kernels are summed here)
        const bool planes = true;      // (every gradient kernel stores one plane per contributing workgroup)
        const int nplanes = k.dense_gemm ? qgdk_dense_sigma_planes(&k) : k.cp / 8;
        HIP_TRY(h, hipMemcpy(out, k.sigma, need * sizeof(double), hipMemcpyDeviceToHost));
        std::vector<double> pl_(need);
        for (int g = 1; planes && g < nplanes; g++) {
            HIP_TRY(h, hipMemcpy(pl_.data(), k.sigma + (size_t)g * need, need * sizeof(double), hipMemcpyDeviceToHost));
            for (size_t e = 0; e < need; e++) out[e] += pl_[e];
        }
        return QGD_OK;
    }
    if (s == "tables") { HIP_TRY(h, hipMemcpy(out, k.tab, need * sizeof(double), hipMemcpyDeviceToHost)); return QGD_OK; }
    memset(out, 0, need * sizeof(double));
    if (s == "Linv") {
        std::vector<double> b(nt * 2 * pl);
        HIP_TRY(h, hipMemcpy(b.data(), k.LinvT, b.size() * sizeof(double), hipMemcpyDeviceToHost));   // row-major planes
        for (size_t n = 1; n < nt; n++) for (size_t r = 0; r < N; r++) for (size_t c = 0; c < N; c++) {
            out[((n * N + r) * N + c) * 2] = b[n * 2 * pl + r * Np + c];
            out[((n * N + r) * N + c) * 2 + 1] = b[n * 2 * pl + pl + r * Np + c];
        }
        return QGD_OK;
    }
    const double *src = (s == "L") ? k.L : (s == "R") ? k.R : k.Pr;
    const size_t cnt = (s == "P") ? nt - 1 : nt;
    std::vector<double> b(cnt * panel);
    HIP_TRY(h, hipMemcpy(b.data(), src, b.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (size_t n = 0; n < cnt; n++) for (size_t r = 0; r < N; r++) for (size_t c = 0; c < N; c++) {
        size_t o = n * panel + panel_index((int)r, (int)c, (int)PW);
        out[((n * N + r) * N + c) * 2] = b[o];
        out[((n * N + r) * N + c) * 2 + 1] = b[o + 8];
    }
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

int qgd_exchange_buffer(qgd_handle h, int32_t which, void **dev_ptr, size_t *total_doubles, size_t *own_offset,
                        size_t *own_doubles)
{
    if (!h || !dev_ptr || !total_doubles || !own_offset || !own_doubles) return QGD_ERR_ARGUMENT;
    NEED_GRID(h);
    const qgdk_ctx &k = h->k;
    const size_t pl = (size_t)k.Np * k.Np, hstep = (size_t)k.Np * 2 * k.cp, W = (size_t)k.part_world;
    if (which == 0) {          // block propagators, all-gather
        const size_t chunk = (size_t)4 * pl;               // [R planes | R panel] of one window
        *dev_ptr = k.RX; *total_doubles = W * chunk; *own_offset = (size_t)k.part_rank * chunk; *own_doubles = chunk;
    } else if (which == 1) {   // affine parts + y_N, all-gather
        const size_t chunk = (size_t)2 * hstep;            // [phi^rank | y_N]
        *dev_ptr = k.phiRX; *total_doubles = W * chunk; *own_offset = (size_t)k.part_rank * chunk; *own_doubles = chunk;
    } else if (which == 2) {   // gradient + scalars, all-reduce(sum)
        if (!k.redbuf) return fail(h, QGD_ERR_STATE, "no control basis set");
        *dev_ptr = k.redbuf; *total_doubles = (size_t)k.n_pcof + 4; *own_offset = 0; *own_doubles = (size_t)k.n_pcof + 4;
    } else if (which == 3) {   // column shards: the three objective scalars at the turnaround, all-reduce(sum)
        if (!k.redbuf) return fail(h, QGD_ERR_STATE, "no control basis set");
        *dev_ptr = k.scal; *total_doubles = 3; *own_offset = 0; *own_doubles = 3;
    } else return fail(h, QGD_ERR_ARGUMENT, "unknown exchange buffer");
    return QGD_OK;
}

int qgd_dist_forward_begin(qgd_handle h, const double *pcof, int32_t n_pcof)
{
    if (!h) return QGD_ERR_ARGUMENT;
    HIP_TRY(h, hipSetDevice(h->device));
    NEED_GRID(h);
    if (!h->have_basis) return fail(h, QGD_ERR_STATE, "qgd_set_control_basis must be called first");
    return forward_begin(h, pcof, n_pcof);
}

int qgd_dist_forward_end(qgd_handle h)
{
    if (!h) return QGD_ERR_ARGUMENT;
    HIP_TRY(h, hipSetDevice(h->device));
    return forward_end(h);
}

int qgd_dist_adjoint_begin(qgd_handle h)
{
    if (!h) return QGD_ERR_ARGUMENT;
    HIP_TRY(h, hipSetDevice(h->device));
    if (!h->k.have_target) return fail(h, QGD_ERR_STATE, "qgd_set_target must be called first");
    if (!h->forward_valid) return fail(h, QGD_ERR_STATE, "no forward evaluation to differentiate");
    return adjoint_begin(h);
}

int qgd_dist_adjoint_end(qgd_handle h)
{
    if (!h) return QGD_ERR_ARGUMENT;
    HIP_TRY(h, hipSetDevice(h->device));
    return adjoint_end(h);
}

int qgd_dist_finish(qgd_handle h, double *grad, double *out3)
{
    if (!h) return QGD_ERR_ARGUMENT;
    HIP_TRY(h, hipSetDevice(h->device));
    return fetch_results(h, grad, out3);
}

// ---------------------------------------------------------------------------
// Multi-GPU inside the library: RCCL over xGMI.  One process (or thread) per GPU, one handle per rank; rank 0 makes
// the 128-byte id, the host moves it to the other ranks by any means it has (MPI.jl, a socket, a file), and every
// rank gives its handle the communicator.  From then on qgd_discrete_adjoint / qgd_eval_forward are COLLECTIVE
// calls: every rank makes them with the same pcof, and the library issues the all-gathers / all-reduces on the
// handle's stream between its own phases.
// ---------------------------------------------------------------------------
int qgd_comm_unique_id(void *id128)
{
    if (!id128) return fail(nullptr, QGD_ERR_ARGUMENT, "null argument");
    RcclApi &R = rccl();
    if (!R.ok) return fail(nullptr, QGD_ERR_COMM, R.err);
    ncclUniqueId id;
    ncclResult_t r = R.GetUniqueId(&id);
    if (r != ncclSuccess) return fail(nullptr, QGD_ERR_COMM, std::string("ncclGetUniqueId: ") + R.GetErrorString(r));
    static_assert(sizeof(id) == QGD_UNIQUE_ID_BYTES, "ncclUniqueId is 128 bytes");
    memcpy(id128, &id, sizeof(id));
    return QGD_OK;
}

int qgd_comm_destroy(qgd_handle h)
{
    if (!h) return QGD_ERR_ARGUMENT;
    if (!h->comm) return QGD_OK;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->k.stream);
    ncclResult_t r = rccl().CommDestroy(h->comm);
    h->comm = nullptr; h->comm_rank = 0; h->comm_world = 1;
    if (r != ncclSuccess) return fail(h, QGD_ERR_COMM, std::string("ncclCommDestroy: ") + rccl().GetErrorString(r));
    return QGD_OK;
}

int qgd_comm_init_rccl(qgd_handle h, const void *unique_id, int32_t rank, int32_t world, int32_t shard)
{
    if (h) drop_graph(h);
    if (!h || !unique_id) return fail(h, QGD_ERR_ARGUMENT, "null argument");
    if (world < 1 || rank < 0 || rank >= world) return fail(h, QGD_ERR_ARGUMENT, "rank/world out of range");
    if (shard != QGD_SHARD_TIME && shard != QGD_SHARD_COLUMNS) return fail(h, QGD_ERR_ARGUMENT, "shard: 0 time windows, 1 column blocks");
    RcclApi &R = rccl();
    if (!R.ok) return fail(h, QGD_ERR_COMM, R.err);
    HIP_TRY(h, hipSetDevice(h->device));
    int rc = qgd_comm_destroy(h);
    if (rc) return rc;
    // time windows: the rank's window of the grid (invalidates control basis and histories, like qgd_set_nsteps);
    // column blocks: the handle was created from the rank's columns, the grid stays whole.
    // A handle with a communicator keeps its window RESIDENT (the collective protocol does not walk the windows of a
    // bounded-memory grid): comm_pending makes this allocation skip the window planner -- a grid that was being
    // processed in windows (memory budget, > 65 000 steps) is allocated whole here, or the call fails with
    // QGD_ERR_MEMORY / QGD_ERR_UNSUPPORTED and the handle keeps the layout it had.
    const int prev_rank = h->part_rank, prev_world = h->part_world;
    const bool deferred = !h->grid_ready;
    auto restore = [&]() {
        h->comm_pending = false;
        h->part_rank = prev_rank; h->part_world = prev_world;
        if (deferred) { free_pool(h->grid_bufs); h->grid_ready = false; }
        else (void)alloc_grid(h);                // (best effort: the caller's error is the one already recorded)
    };
    h->comm_pending = true;
    if (shard == QGD_SHARD_TIME) rc = qgd_set_partition(h, rank, world);
    else if (h->part_world != 1 || h->chunks_eff > 1 || !h->grid_ready) rc = qgd_set_partition(h, 0, 1);
    if (rc) { const std::string e = h->err; const int code = rc; restore(); h->err = e; return code; }
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    ncclComm_t comm = nullptr;
    const ncclResult_t nr = R.CommInitRank(&comm, world, id, rank);
    if (nr != ncclSuccess) {
        restore();
        return fail(h, QGD_ERR_COMM, std::string("ncclCommInitRank: ") + R.GetErrorString(nr));
    }
    h->comm = comm; h->comm_shard = shard; h->comm_rank = rank; h->comm_world = world;
    h->comm_pending = false;
    return QGD_OK;
}

int qgd_set_comm_timeout(qgd_handle h, double milliseconds)
{
    if (!h) return QGD_ERR_ARGUMENT;
    if (!(milliseconds > 0)) return fail(h, QGD_ERR_ARGUMENT, "the time limit of a collective evaluation must be positive (milliseconds)");
    h->comm_timeout_ms = milliseconds;
    return QGD_OK;
}

int qgd_comm_debug_fail_at(qgd_handle h, int32_t collective)
{
    if (!h) return QGD_ERR_ARGUMENT;
    if (collective < 0 || collective > 4) return fail(h, QGD_ERR_ARGUMENT, "collective: 0 off, 1..4 = in front of exchange 0..3");
    h->comm_fail_at = collective;
    return QGD_OK;
}

int qgd_comm_info(qgd_handle h, int32_t *out3)
{
    if (!h || !out3) return QGD_ERR_ARGUMENT;
    out3[0] = h->comm ? h->comm_rank : -1; out3[1] = h->comm ? h->comm_world : 0; out3[2] = h->comm_shard;
    return QGD_OK;
}

// NUMA node the card is attached to (sysfs, by PCI address), -1 when unknown
static int gpu_numa_node(int device)
{
    char bdf[64] = {0};
    if (hipDeviceGetPCIBusId(bdf, (int)sizeof bdf, device) != hipSuccess) { (void)hipGetLastError(); return -1; }
    for (char *q = bdf; *q; q++) *q = (char)tolower((unsigned char)*q);
    const std::string path = std::string("/sys/bus/pci/devices/") + bdf + "/numa_node";
    FILE *f = fopen(path.c_str(), "r");
    if (!f) return -1;
    int node = -1;
    if (fscanf(f, "%d", &node) != 1) node = -1;
    fclose(f);
    return node;
}

int qgd_register_host_buffer(qgd_handle h, void *ptr, size_t bytes)
{
    if (!h || !ptr || !bytes) return fail(h, QGD_ERR_ARGUMENT, "null argument");
    HIP_TRY(h, hipSetDevice(h->device));
    if (find_reg(h, ptr, bytes)) return QGD_OK;
    // QGD_PIN_NUMA=1 (off by default): before the pages of the caller's array are pinned they are moved (and the untouched
    // ones steered) to the NUMA node the card is attached to -- mbind(MPOL_PREFERRED, MPOL_MF_MOVE) on the page-aligned
    // interior of the array -- for hosts whose cross-socket DMA is slow.  On the two-socket MI355X hosts of this pool it makes
    // no difference (0.88 ms for the 31.6 MB of the reference-shaped call with the arrays on either node,
    // scripts/numa_pin_check.py), hence opt-in.
    if (getenv("QGD_PIN_NUMA") && atoi(getenv("QGD_PIN_NUMA")) == 1) {
        const int node = gpu_numa_node(h->device);
        const uintptr_t lo = ((uintptr_t)ptr + 4095) & ~(uintptr_t)4095, hi = ((uintptr_t)ptr + bytes) & ~(uintptr_t)4095;
        if (node >= 0 && node < 64 && hi > lo) {
            unsigned long mask = 1ul << node;
            (void)syscall(SYS_mbind, (void *)lo, (unsigned long)(hi - lo), 1 /* MPOL_PREFERRED */, &mask, 65ul, 2u /* MPOL_MF_MOVE */);
        }
    }
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

// ---------------------------------------------------------------------------
// Column-sharded evaluation: the reference's own parallel axis (Threads.@threads over initial conditions,
// src/forward_evolution.jl:48,332) and the split BASELINE.json's north star sketches.  Every rank's handle is built
// from ITS columns of u0, v0 and the target, with the global N_ess.  Everything is independent per column except the
// overlaps <w_N,R>, <w_N,T> in the terminal condition (global sums, infidelity.jl:13-17): one all-reduce of three
// scalars at the turnaround, one of [grad | scalars] at the end.  The propagator build is replicated on every rank --
// that is why time windows are the default split.
// ---------------------------------------------------------------------------
int qgd_cols_forward(qgd_handle h, const double *pcof, int32_t n_pcof)
{
    if (!h || !pcof) return fail(h, QGD_ERR_ARGUMENT, "null argument");
    HIP_TRY(h, hipSetDevice(h->device));
    NEED_GRID(h);
    if (!h->k.have_target) return fail(h, QGD_ERR_STATE, "qgd_set_target must be called first");
    if (!h->have_basis) return fail(h, QGD_ERR_STATE, "qgd_set_control_basis must be called first");
    return run_forward(h, pcof, n_pcof);        // scal = this rank's <w,R>, <w,T>, guard: exchange buffer 3
}

int qgd_cols_adjoint(qgd_handle h, int32_t keep_scalars)
{
    if (!h) return QGD_ERR_ARGUMENT;
    HIP_TRY(h, hipSetDevice(h->device));
    qgdk_ctx &k = h->k;
    if (!h->forward_valid) return fail(h, QGD_ERR_STATE, "no forward evaluation to differentiate");
    { PhaseTimer t(h, "terminal"); K_TRY(h, qgdk_terminal_given(&k)); }      // y_N from the all-reduced overlaps
    int rc;
    if ((rc = adjoint_begin(h))) return rc;
    if ((rc = adjoint_end(h))) return rc;
    // the final all-reduce sums [grad | scalars]; the scalars are global already: all ranks but one contribute zeros
    if (!keep_scalars) HIP_TRY(h, hipMemsetAsync(k.scal, 0, 3 * sizeof(double), k.stream));
    return QGD_OK;
}

int qgd_set_operator_path(qgd_handle h, int32_t mode)
{
    if (h) drop_graph(h);
    if (!h) return QGD_ERR_ARGUMENT;
    if (mode < 0 || mode > 2) return fail(h, QGD_ERR_ARGUMENT, "operator path: 0 automatic, 1 dense, 2 sparse");
    if (mode == 2 && !h->sparse_available)
        return fail(h, QGD_ERR_UNSUPPORTED, "the operators are too dense (or N > 64) for the sparse kernels");
    h->k.use_sparse = (mode == 2) || (mode == 0 && h->sparse_available && !getenv("QGD_DENSE_OPS"));
    h->forward_valid = false; h->derivs_valid = false;
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
