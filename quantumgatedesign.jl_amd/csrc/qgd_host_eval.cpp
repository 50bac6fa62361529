// qgd_host_eval.cpp -- host side of the C ABI (include/qgd.h), one evaluation: the forward and adjoint phases (orchestration mirrors eval_forward!, src/forward_evolution.jl:33-70, and
// discrete_adjoint!, src/eval_grad_discrete_adjoint.jl:107-160), the transport of results and output arrays, the evaluation entry points.
#include "qgd_host.h"

namespace qgdh {


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
    //  engine path was slow)
    if (!h->copy_stream2) HIP_TRY(h, hipStreamCreateWithFlags(&h->copy_stream2, hipStreamNonBlocking));
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
int download(qgd_handle h, void *dst, const void *src, size_t row_bytes, size_t rows)
{
    if (rows <= 1 || !find_reg(h, dst, row_bytes * rows))
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
// (Writing registered host arrays in place with a few persistent workgroups instead of staging + copying was measured slower
// in round 2 -- 0.96 ms at best against 0.91 for the 31.6 MB of the cnot3 call -- and is gone.)
int copy_history_out(qgd_handle h, double *uv_history, int save)
{
    qgdk_ctx &k = h->k;
    // save > 1 (eval_forward's saveEveryNsteps, forward_evolution.jl:104,178,239-241): slot s of the output holds time
    // point s * save -- the re-layout kernel reads the panels with a stride of `save` time points
    const size_t hstep = (size_t)k.Np * 2 * k.cp * (size_t)save, nt = 1 + ((size_t)k.nt - 1) / (size_t)save, m = k.m, n2 = 2 * (size_t)k.N;
    int rc = copy_side(h);
    if (rc) return rc;
    const long long dcol = (long long)(nt * (m + 1) * n2), dn = (long long)((m + 1) * n2), dj = (long long)n2;
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


// Does this evaluation take the fused front (qgd_front.h)?  A full evaluation on one rank with the grid resident, the control
// basis on the device and pcof small enough for the kernel arguments, a diagonal guard projector or none -- and a grid on which
// it wins: 513 .. 704 time points, where k_front is one round of two to three workgroups per CU and the tail workgroups start
// from pre-built panels (qgdk_front_pre_plan).  Measured (scripts/front_crossover.py, cnot3, us per evaluation front / general):
// 100 steps 185 / 178, 256: 233 / 217, 400: 254 / 256, 512: 284 / 282, 550: 297 / 306, 700: 339 / 342, 800: 430 / 431,
// 1100: 530 / 522 -- without the tail to balance, two launches with eight waves per half time point build faster than four
// waves per time point do.  QGD_PATHS=front takes the front wherever it is supported (tests).
static bool front_applies(qgd_handle h, const double *pcof, int n_pcof)
{
    const qgdk_ctx &k = h->k;
    if (!(pcof && h->have_basis && n_pcof == k.n_pcof && n_pcof <= QGD_PCOF_KERNARG && h->graph_off && !qgd_path("pcof_copy") &&
          !qgd_path("no_front") && !qgd_path("inv_panels") && h->chunks_eff == 1 && h->part_world == 1 && k.part_world == 1 && !h->comm && k.g_nt == 0 &&
          !k.keep_scal && !k.grad_accumulate && k.nt >= 2 && (k.have_guard == 0 || k.have_guard == 2) &&
          (size_t)k.Np * 2 * k.cp < 32768 && k.phi0 && k.hforc && k.termU && k.n_ops > 0 && qgdk_front_supported(&k) != 0)) return false;
    int q2 = 0, q1 = 0;
    return qgd_path("front") != nullptr || qgdk_front_pre_plan(&k, &q2, &q1) > 0;
}


// forward, part 1: everything that needs no other rank (tables .. block propagators)
int forward_begin(qgd_handle h, const double *pcof, int n_pcof, bool allow_front)
{
    qgdk_ctx &k = h->k;
    if (h->stream_dead) return fail(h, QGD_ERR_COMM, "a communicator of this handle was leaked with a collective stuck on its stream: the handle accepts no further evaluation");
    k.front = (allow_front && front_applies(h, pcof, n_pcof)) ? 1 : 0;
    h->front_last = k.front != 0;
    // guard penalty: the guard stage stores its workgroups' partial sums and a later stage adds them in a fixed order (the
    // same bits on every run)
    k.gpart_on = k.gpart ? 1 : 0;
    k.gpart_n = qgdk_guard_parts(&k);
    // who adds the partials up: the terminal stage where this handle runs one right behind the guard stage (the grid
    // resident and the final time its own); a window of a long grid and the other ranks of a partition launch k_guard_fold
    k.gpart_terminal = (h->chunks_eff == 1 && k.part_rank == k.part_world - 1) ? 1 : 0;
    if (pcof) {
        if (!h->have_basis) return fail(h, QGD_ERR_STATE, "qgd_set_control_basis must be called before passing pcof");
        PhaseTimer t(h, "tables");
        if (k.front) {
            K_TRY(h, qgdk_tables_front(&k, pcof, n_pcof));      // tables + the step matrices k_front's workgroups past two per CU start from + phi_0
        } else if (n_pcof == k.n_pcof && n_pcof <= QGD_PCOF_KERNARG && h->graph_off && !qgd_path("pcof_copy")) {   // (a captured graph would freeze the values)
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
        HIP_TRY(h, hipMemsetAsync(k.status, 0, 3 * sizeof(int), k.stream));      // (flag and the two counters; the fourth word is the inverse's memory of the last evaluation)
    }
    if (k.front) {
        { PhaseTimer t(h, "front"); K_TRY(h, qgdk_front(&k)); }      // L_n^-H and S_n = R_n L_n^-1 of every time point, one workgroup each
        { PhaseTimer t(h, "sweep_forward"); K_TRY(h, qgdk_forward_blocks(&k)); }
    } else {
        { PhaseTimer t(h, "build_LR"); K_TRY(h, qgdk_build_LR(&k)); }
        { PhaseTimer t(h, "inverse"); K_TRY(h, qgdk_inverse(&k)); }
        if (qgdk_propagator_is_fused(&k)) { K_TRY(h, qgdk_propagator(&k)); }   // k_inverse_mfma formed P_n already
        else { PhaseTimer t(h, "propagator"); K_TRY(h, qgdk_propagator(&k)); }
        { PhaseTimer t(h, "sweep_forward"); K_TRY(h, qgdk_forward_blocks(&k)); }
    }
    h->forward_valid = false; general_history(h);
    h->derivs_valid = false;
    h->status_dirty = true;       // (a general-path evaluation can leave the status word set without its result passing through fail():
                                  //  qgd_cols_forward, qgd_dist_* without qgd_dist_finish -- the small-problem path clears it from the host)
    return QGD_OK;
}


// forward, part 2: after the block propagators of all ranks are in PiX
int forward_end(qgd_handle h)
{
    qgdk_ctx &k = h->k;
    { PhaseTimer t(h, "sweep_forward2"); K_TRY(h, qgdk_forward_finish(&k)); }
    if (k.front) {
        // the sweep ran in phi = L psi: state history, guard forcing and penalty, the adjoint sweep's forcing L^-H f
        if (k.have_guard == 0 && !h->forcing_zero) {
            HIP_TRY(h, hipMemsetAsync(k.forcing, 0, (size_t)k.nt * k.Np * 2 * k.cp * sizeof(double), k.stream));
            h->forcing_zero = true;
        }
        if (k.have_guard) h->forcing_zero = false;
        PhaseTimer t(h, "psi"); K_TRY(h, qgdk_psi(&k));
    } else if (k.have_guard == 0 && h->forcing_zero) {
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
    if (!k.front) { PhaseTimer t(h, "lambda"); K_TRY(h, qgdk_lambda(&k)); }      // (fused front: the sweep ran in lambda itself)
    if (h->lambda_out) {      // its download runs beside the gradient kernels
        double *out = h->lambda_out; h->lambda_out = nullptr;
        int rc = h->lambda_derivs ? copy_lambda_full_out(h, out) : copy_panels_out(h, k.lam, &h->stage_lam, out, (size_t)k.m + 1, 1);
        if (rc) return rc;
    }
    if (!h->derivs_valid && qgdk_gradient_needs_derivs(&k)) { PhaseTimer t(h, "derivs"); K_TRY(h, qgdk_derivs(&k)); h->derivs_valid = true; }
    { PhaseTimer t(h, "gradient"); K_TRY(h, qgdk_gradient(&k)); }
    return QGD_OK;
}


int run_forward(qgd_handle h, const double *pcof, int n_pcof, bool allow_front)
{
    if (h->chunks_eff > 1) return chunked_forward(h, pcof, n_pcof);
    if (h->part_world != 1) return fail(h, QGD_ERR_STATE, "partitioned handle: use the qgd_dist_* entry points");
    int rc = forward_begin(h, pcof, n_pcof, allow_front);
    if (rc) return rc;
    if ((rc = forward_end(h))) return rc;
    if (pcof) h->fwd_pcof.assign(pcof, pcof + n_pcof); else h->fwd_pcof.clear();
    h->history_stale = false;
    return QGD_OK;
}



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
    if (h->status_dirty) { HIP_TRY(h, hipMemsetAsync(k.status, 0, 3 * sizeof(int), k.stream)); h->status_dirty = false; }
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



// status + results of an evaluation in one device-to-host copy (the three separate copies cost
// ~25 us of the 0.5 ms evaluation on cnot3)
int fetch_results(qgd_handle h, double *grad, double *out3, const double *src)
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


bool same_pcof(qgd_handle h, const double *pcof, int n_pcof)
{
    if (h->history_stale) return false;     // (the last evaluation ran on the small-problem path: no stored history to reuse)
    return pcof ? ((size_t)n_pcof == h->fwd_pcof.size() && n_pcof > 0 && !memcmp(pcof, h->fwd_pcof.data(), sizeof(double) * n_pcof))
                : h->fwd_pcof.empty();
}

}  // namespace qgdh

using namespace qgdh;

extern "C" {


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
    int rc = run_forward(h, pcof, n_pcof, true);
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
            h->forward_valid = true; general_history(h);
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
        rc = run_forward(h, pcof, n_pcof, true);
        if (rc) { h->defer_terminal = false; return rc; }
    }
    // The downloads run on the copy stream beside the adjoint sweep and are what bounds this form of the call (PCIe):
    // the guard forcing goes first -- it is final once the forward sweep is (eval_grad_discrete_adjoint.jl:732-752) and
    // keeps the link busy while the stage derivatives of the state history are still being computed and laid out
    if (adjoint_forcing && (rc = copy_panels_out(h, k.forcing, &h->stage_f, adjoint_forcing, 1, 0))) return rc;
    if (uv_history) {
        if (!h->derivs_valid) { PhaseTimer t(h, "derivs"); K_TRY(h, qgdk_derivs(&k)); h->derivs_valid = true; }
        if ((rc = copy_history_out(h, uv_history))) return rc;
    }
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
    if ((rc = fetch_results(h, grad, out3))) return rc;
    rc = finish_copies(h);
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
    h->forward_valid = false; general_history(h);      // this history is not the one the adjoint sweep differentiates
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
        h->forward_valid = false; general_history(h);
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
    h->forward_valid = false; general_history(h);   // the state history was not computed
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
    else if (s == "front_path") need = 1;
    else return fail(h, QGD_ERR_ARGUMENT, "unknown intermediate '" + s + "'");
    if (needed) *needed = need;
    if (!out) return QGD_OK;
    if (capacity < need) return fail(h, QGD_ERR_ARGUMENT, "buffer too small");
    if (s == "small_path") { out[0] = h->history_stale ? 1.0 : 0.0; return QGD_OK; }
    if (s == "front_path") { out[0] = h->front_last ? 1.0 : 0.0; return QGD_OK; }         // did the LAST forward evaluation take the fused front      // did the LAST evaluation run on the small-problem path
    if (s == "selection") {      // which kernel families this problem runs on (tests assert that a shape selects what it is meant to)
        out[0] = k.use_sparse ? 2.0 : (k.dense_gemm ? 1.0 : 0.0);      // 2 sparse (ELL), 1 N > 64 GEMM-style kernels, 0 dense N <= 64
        out[1] = (k.dense_gemm && !k.use_sparse) ? (double)qgdk_dense_sigma_form(&k) : -1.0;
        out[2] = (k.dense_gemm && k.binv) ? 1.0 : 0.0;               // block Gauss-Jordan inverse over 64-column blocks
        out[3] = (double)h->chunks_eff;
        return QGD_OK;
    }
    NEEDS_RESIDENT_GRID(h, "qgd_get_intermediate");
    if (h->stream_dead) return fail(h, QGD_ERR_COMM, "a communicator of this handle was leaked with a collective stuck on its stream");
    if (h->history_stale && s != "repivoted" && !h->tiny_pcof.empty()) {
        // the last evaluation ran on the small-problem path, which keeps no intermediates: the same evaluation once more on
        // the general path (diagnostics only)
        const std::vector<double> pc = h->tiny_pcof;
        int rcs = run_forward(h, pc.data(), (int)pc.size());
        if (!rcs && h->tiny_was_gradient && k.have_target) { rcs = adjoint_begin(h); if (!rcs) rcs = adjoint_end(h); }
        if (rcs) return rcs;
    }
    if (h->front_last && (s == "L" || s == "R" || s == "Linv" || s == "P") && !h->fwd_pcof.empty()) {
        // the last evaluation took the fused front, whose step matrices are the same-point form's (L^H, R^H, L^-H, S): the
        // forward evaluation once more on the general path (diagnostics only)
        const std::vector<double> pc = h->fwd_pcof;
        const int rcs = run_forward(h, pc.data(), (int)pc.size(), false);
        if (rcs) return rcs;
    }
    HIP_TRY(h, hipStreamSynchronize(k.stream));
    if (s == "repivoted") {     // last inverse launch: matrices not done by the first attempt + 65536 * (N = 64: matrices that went on to the
                                // fully pivoted elimination); qgd_inverse_cb.h, qgd_k_dense.hip
        int v[3] = {0, 0, 0};
        HIP_TRY(h, hipMemcpy(v, k.status, 3 * sizeof(int), hipMemcpyDeviceToHost));
        out[0] = (double)v[1] + 65536.0 * (double)v[2];
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
    if (!h->forward_valid || h->history_stale) return fail(h, QGD_ERR_STATE, "no forward evaluation to differentiate (qgd_dist_forward_* first)");
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
    if (!h->forward_valid || h->history_stale) return fail(h, QGD_ERR_STATE, "no forward evaluation to differentiate (qgd_cols_forward first)");
    { PhaseTimer t(h, "terminal"); K_TRY(h, qgdk_terminal_given(&k)); }      // y_N from the all-reduced overlaps
    int rc;
    if ((rc = adjoint_begin(h))) return rc;
    if ((rc = adjoint_end(h))) return rc;
    // the final all-reduce sums [grad | scalars]; the scalars are global already: all ranks but one contribute zeros
    if (!keep_scalars) HIP_TRY(h, hipMemsetAsync(k.scal, 0, 3 * sizeof(double), k.stream));
    return QGD_OK;
}

}  // extern "C"
