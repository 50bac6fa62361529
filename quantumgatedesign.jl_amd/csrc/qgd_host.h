// qgd_host.h -- what the four host-side translation units of the C ABI share (include/qgd.h): the handle, the error and
// launch macros, the phase timer and the internal entry points of each unit.
//   qgd_host_alloc.cpp    handles: creation, validation (SchrodingerProb.jl:73-154), the time grid and its windows, setters
//   qgd_host_eval.cpp     one evaluation: forward / adjoint phases, result transport, the evaluation entry points
//   qgd_host_windows.cpp  time grids in bounded memory: the windowed forward, adjoint and forced sweeps
//   qgd_host_comm.cpp     several GPUs: the RCCL binding, the collective evaluation and its failure mode
// There is no CPU fallback: without a GPU every compute entry point fails.
#pragma once
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

namespace qgdh {

extern thread_local std::string g_create_error;

struct Phase { const char *name; int slot; hipEvent_t e0, e1; bool used; };

}  // namespace qgdh
using qgdh::Phase;

struct qgd_handle_s {
    qgdk_ctx k{};
    int order = 0, nsteps = 0, device = 0;      // nsteps: GLOBAL number of timesteps
    int part_rank = 0, part_world = 1;
    bool own_stream = true;
    bool timing = false;                // per-phase HIP events are opt-in (qgd_set_timing): 26 event records cost ~0.17 ms
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
    // The fused front (qgd_device.h: qgdk_ctx::front) takes full evaluations of qgd_eval_forward / qgd_discrete_adjoint on problems
    // qgdk_front_supported admits.  front_last: the device buffers hold such an evaluation -- state history, lambda and forcing
    // are the same quantities as ever, but L / R / Linv / P are the same-point form's: an entry point that wants the two-point
    // form's (qgd_get_intermediate) redoes the forward evaluation on the general path first.
    bool front_last = false;
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
    hipStream_t copy_stream2 = nullptr;   // the second half of a large pinned download goes to a second DMA engine 
    hipEvent_t ev_ready = nullptr;
    std::vector<double> fwd_pcof;       // the pcof of the forward sweep that is on the device (history_precomputed)
    std::vector<double> scatter_tmp;    // unregistered lambda_history: compact copy, scattered on the host
    bool copies_pending = false;
    bool forcing_zero = false;          // no guard projector: the adjoint forcing is all zeros and nothing has written it since
                                        // (alloc_grid clears it, qgd_eval_adjoint uploads a caller's forcing into it)
    bool defer_terminal = false;        // a full gradient evaluation: the overlaps and y_N ride in the first adjoint launch
    double *lambda_out = nullptr;       // lambda_history of the evaluation in flight (copied out right after the lambda phase)
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
    double comm_timeout_ms = 30000.0;
    bool stream_dead = false;           // a communicator had to be leaked with a collective stuck on the stream (no ncclCommAbort in this librccl):
                                        // every later compute entry point fails fast with QGD_ERR_COMM, qgd_destroy does not wait for the stream
    int comm_fail_at = 0;               // != 0: pretend a local failure in front of collective #n (set only by the tests' fault injector, tests/hooks/qgd_test_hooks.cpp -- no entry point of the library writes it)
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

namespace qgdh {

int fail(qgd_handle h, int code, const std::string &msg);


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


inline double factorial(int n) { double f = 1; for (int i = 2; i <= n; i++) f *= i; return f; }

// hermite.jl:389-391
inline double hermite_coefficient(int j, int p, int q) { return factorial(p) * factorial(p + q - j) / (factorial(p + q) * factorial(p - j)); }


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
// A history of the GENERAL path is about to be produced, or the stored one is void: whatever the small-problem path left behind
// (history_stale: "the device buffers hold no history of the last evaluation"; tiny_pcof: the pcof to redo it from) no longer
// describes the handle.
inline void general_history(qgd_handle h) { h->history_stale = false; h->tiny_pcof.clear(); }


// (QGD_CREATE_DEFER_GRID) entry points that read or size anything by the time grid allocate it first
#define NEED_GRID(h)                                                                           \
    do { if (!(h)->grid_ready) { int rg__ = alloc_grid(h); if (rg__) return rg__; } } while (0)


#define K_TRY(h, expr)                                                                         \
    do {                                                                                       \
        int e__ = (expr);                                                                      \
        if (e__ != 0)                                                                          \
            return fail((h), QGD_ERR_NO_DEVICE, std::string(#expr) + ": " + hipGetErrorString((hipError_t)e__)); \
    } while (0)


#define NEEDS_RESIDENT_GRID(h, what)                                                                                   \
    do { if ((h)->chunks_eff > 1) return fail((h), QGD_ERR_UNSUPPORTED, std::string(what) + " needs the whole time grid resident: this handle " \
                                               "processes it in " + std::to_string((h)->chunks_eff) + " windows (raise qgd_set_memory_budget)"); } while (0)



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


#define NCCL_TRY(h, expr)                                                                      \
    do {                                                                                       \
        ncclResult_t r__ = (expr);                                                             \
        if (r__ != ncclSuccess)                                                                \
            return fail((h), QGD_ERR_COMM, std::string(#expr) + ": " + rccl().GetErrorString(r__)); \
    } while (0)

RcclApi &rccl();

// qgd_host_alloc.cpp
void free_pool(std::vector<void *> &pool);
void drop_graph(qgd_handle h);
int plan_windows(qgd_handle h, int chunks, int win);
int alloc_grid(qgd_handle h);

// qgd_host_eval.cpp
qgd_handle_s::HostReg *find_reg(qgd_handle h, const void *p, size_t bytes);
int copy_side(qgd_handle h);
int hand_over(qgd_handle h);
int finish_copies(qgd_handle h);
int download(qgd_handle h, void *dst, const void *src, size_t row_bytes, size_t rows);
int copy_history_out(qgd_handle h, double *uv_history, int save = 1);
int copy_panels_out(qgd_handle h, const double *panels, double **stage, double *out, size_t J, int n_first);
int copy_lambda_full_out(qgd_handle h, double *out);
int upload_pcof(qgd_handle h, const double *pcof, int n_pcof);
int forward_begin(qgd_handle h, const double *pcof, int n_pcof, bool allow_front = false);
int forward_end(qgd_handle h);
int adjoint_begin(qgd_handle h);
int adjoint_end(qgd_handle h);

// qgd_host_windows.cpp
int window_history_out(qgd_handle h, double *uv_history, int save = 1);
int window_lambda_full_out(qgd_handle h, double *out);
int window_panels_out(qgd_handle h, const double *panels, double **stage, double *out, size_t J, int n_first);
int window_tables(qgd_handle h);
int chunk_forward(qgd_handle h, const double *pcof, int n_pcof, int r, bool rerun);
int chunked_forward(qgd_handle h, const double *pcof, int n_pcof, double *uv_history = nullptr, int save = 1);
int chunked_adjoint(qgd_handle h, double *lambda_history = nullptr, double *adjoint_forcing = nullptr);
int chunked_eval_adjoint(qgd_handle h, const double *pcof, int n_pcof, const double *terminal_condition, const double *forcing, double *lambda_history);
int forcing_buffers(qgd_handle h, size_t nt, size_t B);
int upload_forcing(qgd_handle h, const double *forcing, size_t nt, size_t n_off);
int chunked_forward_forced(qgd_handle h, const double *pcof, int n_pcof, const double *forcing, double *uv_history, double *out3);

// qgd_host_eval.cpp
int run_forward(qgd_handle h, const double *pcof, int n_pcof, bool allow_front = false);
bool tiny_applies(qgd_handle h, const double *pcof, int n_pcof);
int tiny_evaluate(qgd_handle h, const double *pcof, int n_pcof, bool gradient, double *grad, double *out3);
int check_status(qgd_handle h);
int fetch_results(qgd_handle h, double *grad, double *out3, const double *src = nullptr);

// qgd_host_comm.cpp
RcclApi load_rccl();
bool comm_abort(qgd_handle h);
int comm_failed(qgd_handle h, const std::string &why);
int comm_local_error(qgd_handle h, int rc);
int comm_wait(qgd_handle h);
int comm_collective(qgd_handle h, int which);
const double *comm_result(qgd_handle h);

// qgd_host_eval.cpp
bool same_pcof(qgd_handle h, const double *pcof, int n_pcof);

// qgd_host_comm.cpp
int comm_forward(qgd_handle h, const double *pcof, int n_pcof);
int comm_discrete_adjoint(qgd_handle h, const double *pcof, int n_pcof, int history_precomputed, double *grad, double *uv_history, double *lambda_history, double *adjoint_forcing, double *out3);
int comm_discrete_adjoint_body(qgd_handle h, const double *pcof, int n_pcof, int history_precomputed, double *grad, double *uv_history, double *lambda_history, double *adjoint_forcing, double *out3);
int comm_eval_forward(qgd_handle h, const double *pcof, int n_pcof, double *uv_history, double *out3);
int comm_eval_forward_body(qgd_handle h, const double *pcof, int n_pcof, double *uv_history, double *out3);

}  // namespace qgdh
