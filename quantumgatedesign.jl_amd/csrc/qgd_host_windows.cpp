// qgd_host_windows.cpp -- host side of the C ABI (include/qgd.h), time grids in bounded memory (qgd_set_memory_budget): the windowed forward, adjoint and forced sweeps.
#include "qgd_host.h"

namespace qgdh {


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
int window_history_out(qgd_handle h, double *uv_history, int save)       // [2N, 1+m, 1 + (nt_glob-1)/save, c]
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


int chunked_forward(qgd_handle h, const double *pcof, int n_pcof, double *uv_history, int save)
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


int chunked_adjoint(qgd_handle h, double *lambda_history, double *adjoint_forcing)
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
    h->forward_valid = false; general_history(h); h->resident_window = -1;      // (the buffers will hold no window's forward history)
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
    h->forward_valid = false; general_history(h); h->resident_window = -1;       // (the window-boundary states are those of the FORCED sweep from here on)
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

}  // namespace qgdh

using namespace qgdh;

extern "C" {

}  // extern "C"
