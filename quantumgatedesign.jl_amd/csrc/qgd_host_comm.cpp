// qgd_host_comm.cpp -- host side of the C ABI (include/qgd.h), several GPUs inside the library: the RCCL binding (dlopen), the collective evaluation and its failure mode.
#include "qgd_host.h"

namespace qgdh {


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


// ---------------------------------------------------------------------------
// Failure mode of the collective calls.  A collective that one rank never enters blocks the others inside an RCCL
// kernel for good, so (1) the single host wait of a collective evaluation is a bounded hipStreamQuery loop that also
// polls ncclCommGetAsyncError, and (2) a rank that fails locally between two collectives, sees an asynchronous RCCL
// error or runs out of time ABORTS its communicator (ncclCommAbort makes the RCCL kernels on the stream return) and
// reports QGD_ERR_COMM.  The handle is left without a communicator: the host tears the job down (bench.py: the rank
// process exits non-zero) or builds a fresh communicator.  There is no retry inside the library.
// ---------------------------------------------------------------------------
// Returns false when the communicator could only be LEAKED (a librccl without ncclCommAbort): the stuck collective is then
// still on the handle's stream, and the handle must not wait for that stream again.
bool comm_abort(qgd_handle h)
{
    if (!h->comm) return true;
    RcclApi &R = rccl();
    ncclComm_t c = h->comm;
    h->comm = nullptr; h->comm_rank = 0; h->comm_world = 1;
    if (!R.CommAbort) {
        // A build of the library without ncclCommAbort: destroying a communicator with a collective stuck on the stream
        // blocks, and so does a wait for the stream -- the bounded-wait promise of qgd.h would not hold.  The communicator
        // is leaked instead and the handle's stream is left alone; the call returns QGD_ERR_COMM.
        (void)hipGetLastError();
        h->stream_dead = true;
        return false;
    }
    (void)R.CommAbort(c);
    (void)hipStreamSynchronize(h->k.stream);      // the library's own kernels behind the aborted collective drain normally
    (void)hipGetLastError();
    return true;
}


int comm_failed(qgd_handle h, const std::string &why)
{
    if (comm_abort(h)) return fail(h, QGD_ERR_COMM, why + "; the communicator of this handle was aborted (ncclCommAbort)");
    return fail(h, QGD_ERR_COMM, why + "; this librccl has no ncclCommAbort: the communicator was leaked with its collective still on the handle's "
                                       "stream -- the handle accepts no further evaluation (destroy it; its stream is not waited for)");
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
                                      " ms (qgd_set_comm_timeout): another rank failed or never made the call");
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
        return fail(h, QGD_ERR_NO_DEVICE, "injected failure in front of collective " + std::to_string(which) + " (tests/hooks/qgd_test_hooks.cpp)");
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

}  // namespace qgdh

using namespace qgdh;

extern "C" {


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


int qgd_comm_info(qgd_handle h, int32_t *out3)
{
    if (!h || !out3) return QGD_ERR_ARGUMENT;
    out3[0] = h->comm ? h->comm_rank : -1; out3[1] = h->comm ? h->comm_world : 0; out3[2] = h->comm_shard;
    return QGD_OK;
}

}  // extern "C"
