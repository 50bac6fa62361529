"""GPU tests of the bounded-memory time grid (qgd_set_memory_budget): a grid whose step matrices
do not fit the budget is processed in windows that share one set of buffers -- forward pass over the windows, adjoint
pass back over them with the window's matrices formed again.  Same discrete quantities in the same order inside a
window, so the results must equal the resident evaluation to rounding."""
import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu


def _pair(qgd, prob, ctrl, pcof, target, order, budget):
    dp = qgd.DeviceProblem(prob, order); dp.set_controls(ctrl); dp.set_target(target)
    assert dp.memory_plan()["windows"] == 1
    g_ref, o_ref = dp.discrete_adjoint(pcof)
    f_ref = dp.eval_forward(pcof)
    dp.close()
    dp = qgd.DeviceProblem(prob, order)
    dp.set_memory_budget(budget)
    dp.set_controls(ctrl); dp.set_target(target)
    return dp, g_ref, np.asarray(o_ref), np.asarray(f_ref)


@pytest.mark.parametrize("which,order,nsteps,windows", [("cnot3", 8, 550, 4), ("cnot3", 8, 137, 3), ("cnot2", 8, 100, 3), ("guarded", 6, 90, 2),
                                                        ("dense_guard", 6, 48, 3), ("synthetic", 12, 240, 3), ("synthetic", 4, 301, 4),
                                                        ("synthetic_square", 8, 48, 3)])
def test_chunked_grid_matches_resident(qgd, which, order, nsteps, windows):
    if which == "synthetic":
        prob, ctrl, pcof, target = cases.synthetic_case(qgd, N=80, c=16, nsteps=nsteps, tf=0.002 * nsteps)
    elif which == "synthetic_square":      # as many columns as rows: the gradient on N x N matrices, the full-chip chain steps,
        prob, ctrl, pcof, target = cases.synthetic_case(qgd, N=176, c=176, n_ops=2, nsteps=nsteps, tf=0.002 * nsteps)   # three blocks in the inverse
    else:
        prob, ctrl, pcof, target = getattr(cases, which + "_case")(qgd, nsteps=nsteps, tf=float(nsteps) / (1 if which.startswith("cnot") else 2))
    probe = qgd.DeviceProblem(prob, order)
    full = probe.memory_plan()["window_bytes"]
    probe.close()
    dp, g_ref, o_ref, f_ref = _pair(qgd, prob, ctrl, pcof, target, order, int(full / windows * 1.15))
    plan = dp.memory_plan()
    # (N > 64 has ~95 MB of buffers that do not shrink with the window -- inverse work slabs -- hence the loose upper bound)
    assert 2 <= plan["windows"] <= 4 * windows and plan["window_bytes"] <= plan["budget"], plan
    scale = max(1.0, np.abs(o_ref).max())
    reps = []
    for rep in range(2):
        g, o = dp.discrete_adjoint(pcof)
        assert np.abs(g - g_ref).max() <= 1e-11 * np.abs(g_ref).max(), (which, plan, rep)
        assert np.abs(np.asarray(o) - o_ref).max() <= 1e-12 * scale
        reps.append((g, np.asarray(o)))
    if which in ("cnot3", "cnot2", "guarded"):       # sparse-operator kernels: windows too give the same bits on every run, guard sum included
        assert np.array_equal(reps[0][0], reps[1][0]) and np.array_equal(reps[0][1], reps[1][1])
    f = dp.eval_forward(pcof)
    assert np.abs(np.asarray(f) - f_ref).max() <= 1e-12 * scale
    g, _ = dp.discrete_adjoint(pcof, history_precomputed=True)          # reuses the stored window-boundary states
    assert np.abs(g - g_ref).max() <= 1e-11 * np.abs(g_ref).max()
    g2, _ = dp.discrete_adjoint(0.5 * pcof, history_precomputed=True)    # a different pcof: the sweep is redone
    chk = qgd.DeviceProblem(prob, order); chk.set_controls(ctrl); chk.set_target(target)
    g2_ref, _ = chk.discrete_adjoint(0.5 * pcof)
    chk.close()
    assert np.abs(g2 - g2_ref).max() <= 1e-11 * np.abs(g2_ref).max()
    # the reference-shaped call: the three arrays are filled window by window and equal the resident call's
    shape = (prob.real_system_size, 1 + order // 2, 1 + prob.nsteps, prob.N_initial_conditions)
    chk = qgd.DeviceProblem(prob, order); chk.set_controls(ctrl); chk.set_target(target)
    ref = [np.zeros(shape, order="F"), np.zeros(shape, order="F"), np.zeros((shape[0], shape[2], shape[3]), order="F")]
    chk.discrete_adjoint(pcof, False, *ref)
    chk.close()
    got = [np.full(shape, np.nan, order="F"), np.full(shape, np.nan, order="F"), np.full((shape[0], shape[2], shape[3]), np.nan, order="F")]
    g, _ = dp.discrete_adjoint(pcof, False, *got)
    assert np.abs(g - g_ref).max() <= 1e-11 * np.abs(g_ref).max()
    for name, a, b in zip(("uv_history", "lambda_history", "adjoint_forcing"), got, ref):
        for j in range(a.shape[1] if a.ndim == 4 else 1):      # per Taylor index: the high coefficients of a random problem are large
            x, y = (a[:, j], b[:, j]) if a.ndim == 4 else (a, b)
            assert np.isfinite(x).all() and np.abs(x - y).max() <= 1e-11 * max(1.0, np.abs(y).max()), (name, j)
    hist = np.full(shape, np.nan, order="F")
    dp.eval_forward(pcof, hist)
    assert np.abs(hist[:, 0] - ref[0][:, 0]).max() <= 1e-11
    # the derivative columns of lambda_history (forward_evolution.jl:427-433) window by window: equal to the resident call's
    chk = qgd.DeviceProblem(prob, order); chk.set_controls(ctrl); chk.set_target(target)
    chk.set_lambda_derivatives(True)
    lam_ref = np.zeros(shape, order="F")
    chk.discrete_adjoint(pcof, False, None, lam_ref, None)
    dp.set_lambda_derivatives(True)
    lam = np.full(shape, np.nan, order="F")
    dp.discrete_adjoint(pcof, False, None, lam, None)
    for j in range(shape[1]):
        assert np.isfinite(lam[:, j]).all() and np.abs(lam[:, j] - lam_ref[:, j]).max() <= 1e-11 * max(1.0, np.abs(lam_ref[:, j]).max()), ("lambda derivatives", j)
    assert np.abs(lam_ref[:, 1:, 1:]).max() > 0 and np.abs(lam[:, :, 0]).max() == 0      # (filled; time index 0 never written)
    dp.set_lambda_derivatives(False)
    # eval_forward's saveEveryNsteps (forward_evolution.jl:104,239-241) window by window: slot s = global time point s * save,
    # whatever the windows' boundaries are (strides that divide the window length, that do not, and one longer than a window)
    for save in (2, 3, 7, plan["steps_per_window"] + 1):
        if save > prob.nsteps:
            continue
        slots = 1 + prob.nsteps // save
        chk.set_save_every(save); dp.set_save_every(save)
        h_ref = np.zeros((shape[0], shape[1], slots, shape[3]), order="F")
        chk.eval_forward(pcof, h_ref)
        h_win = np.full_like(h_ref, np.nan, order="F")
        dp.eval_forward(pcof, h_win)
        for j in range(shape[1]):
            assert np.isfinite(h_win[:, j]).all() and np.abs(h_win[:, j] - h_ref[:, j]).max() <= 1e-11 * max(1.0, np.abs(h_ref[:, j]).max()), ("save", save, j)
        assert np.abs(h_ref[:, 0, 1] - ref[0][:, 0, save]).max() <= 1e-11
    chk.set_save_every(1); dp.set_save_every(1)
    chk.close()
    # eval_adjoint (forward_evolution.jl:300-315, :352-483) window by window: a given terminal condition and a given forcing,
    # with and without the derivative columns -- equal to the resident call
    rng = np.random.default_rng(7)
    term = rng.standard_normal((shape[0], shape[3]))
    forc = rng.standard_normal((shape[0], shape[2], shape[3])) * 0.1
    chk = qgd.DeviceProblem(prob, order); chk.set_controls(ctrl)
    for derivs in (False, True):
        chk.set_lambda_derivatives(derivs); dp.set_lambda_derivatives(derivs)
        for fo in (None, forc):
            l_ref = chk.eval_adjoint(pcof, term, fo)
            l_win = dp.eval_adjoint(pcof, term, fo)
            for j in range(shape[1]):
                assert np.isfinite(l_win[:, j]).all() and np.abs(l_win[:, j] - l_ref[:, j]).max() <= 1e-11 * max(1.0, np.abs(l_ref[:, j]).max()), ("eval_adjoint", derivs, fo is None, j)
            assert np.array_equal(l_win[:, 0, -1], term) and np.abs(l_win[:, :, 0]).max() == 0
    chk.set_lambda_derivatives(False); dp.set_lambda_derivatives(False)
    chk.close()
    g, _ = dp.discrete_adjoint(pcof)                     # (the windows' buffers were reused: the next evaluation starts over)
    assert np.abs(g - g_ref).max() <= 1e-11 * np.abs(g_ref).max()
    # eval_forward with a forcing (forward_evolution.jl:118-129,167-206) window by window: scalars (the guard penalty sums
    # over the windows) and the history with its stage derivatives, equal to the resident call's
    m = order // 2
    ff = np.asfortranarray(0.2 * rng.standard_normal((shape[0], m, shape[2], shape[3])))
    chk = qgd.DeviceProblem(prob, order); chk.set_controls(ctrl); chk.set_target(target)
    h_ref = np.zeros(shape, order="F"); h_win = np.full(shape, np.nan, order="F")
    s_ref = np.asarray(chk.eval_forward_forced(pcof, ff, h_ref))
    s_win = np.asarray(dp.eval_forward_forced(pcof, ff, h_win))
    assert np.abs(s_win - s_ref).max() <= 1e-11 * max(1.0, np.abs(s_ref).max()), (s_win, s_ref)
    for j in range(shape[1]):
        assert np.isfinite(h_win[:, j]).all() and np.abs(h_win[:, j] - h_ref[:, j]).max() <= 1e-11 * max(1.0, np.abs(h_ref[:, j]).max()), ("forced", j)
    s_win2 = np.asarray(dp.eval_forward_forced(pcof, ff))          # (without the history)
    assert np.abs(s_win2 - s_ref).max() <= 1e-11 * max(1.0, np.abs(s_ref).max())
    chk.close()
    g, _ = dp.discrete_adjoint(pcof)                     # (and the ordinary evaluation afterwards starts over)
    assert np.abs(g - g_ref).max() <= 1e-11 * np.abs(g_ref).max()
    # the forced gradient (eval_grad_forced.jl:17-194) window by window: the sensitivities of all parameters continue from
    # window to window, the guard part accumulates -- equal to the resident call's and to the adjoint gradient
    if len(pcof) <= 200 and prob.N_initial_conditions * len(pcof) <= 4096:
        chk = qgd.DeviceProblem(prob, order); chk.set_controls(ctrl); chk.set_target(target)
        gf_ref = chk.eval_grad_forced(pcof)
        chk.close()
        gf = dp.eval_grad_forced(pcof)
        assert np.abs(gf - gf_ref).max() <= 1e-10 * np.abs(gf_ref).max(), np.abs(gf - gf_ref).max() / np.abs(gf_ref).max()
        assert np.abs(gf - g_ref).max() <= 1e-9 * np.abs(g_ref).max()
        g, _ = dp.discrete_adjoint(pcof)
        assert np.abs(g - g_ref).max() <= 1e-11 * np.abs(g_ref).max()
    # what still needs the grid resident says so
    with pytest.raises(qgd._lib.QGDError) as e:
        dp.intermediate("P")
    assert e.value.code == qgd._lib.QGD_ERR_UNSUPPORTED
    dp.close()


def test_budget_too_small_is_a_memory_error(qgd):
    prob, ctrl, pcof, target = cases.cnot3_case(qgd, nsteps=40, tf=40.0)
    dp = qgd.DeviceProblem(prob, 8)
    with pytest.raises(qgd._lib.QGDError) as e:
        dp.set_memory_budget(100_000)           # less than one time step's matrices
    assert e.value.code == qgd._lib.QGD_ERR_MEMORY
    dp.close()


def test_reference_low_order_long_grid(qgd):
    """examples/cnot3_optimize_gate.sb:27-40 runs cnot3 at order 2 with dt = 1e-2: 55 000 time steps over tf = 550.
    Resident that is 25 GB of step matrices; under a 2 GB budget the grid takes ~13 windows.  Same gradient."""
    import time
    nsteps = 55_000
    prob, ctrl, pcof, target = cases.cnot3_case(qgd, nsteps=nsteps, tf=550.0)
    order = 2
    out = {}
    for label, budget in (("resident", 0), ("2GB", 2 << 30)):
        dp = qgd.DeviceProblem(prob, order)
        if budget:
            dp.set_memory_budget(budget)
        dp.set_controls(ctrl); dp.set_target(target)
        plan = dp.memory_plan()
        dp.discrete_adjoint(pcof)
        t0 = time.perf_counter()
        g, o = dp.discrete_adjoint(pcof)
        out[label] = (g, np.asarray(o), plan, time.perf_counter() - t0)
        dp.close()
    assert out["resident"][2]["windows"] == 1 and out["2GB"][2]["windows"] >= 8, (out["resident"][2], out["2GB"][2])
    assert out["2GB"][2]["window_bytes"] <= 2 << 30
    g_ref = out["resident"][0]
    assert np.abs(out["2GB"][0] - g_ref).max() <= 1e-11 * np.abs(g_ref).max()
    assert np.abs(out["2GB"][1] - out["resident"][1]).max() <= 1e-11
    print(f"\n55000-step cnot3, order 2: resident {out['resident'][3] * 1e3:.1f} ms ({out['resident'][2]['window_bytes'] / 2**30:.1f} GiB), "
          f"2 GB budget {out['2GB'][3] * 1e3:.1f} ms in {out['2GB'][2]['windows']} windows")


def test_grid_longer_than_one_launch_dimension(qgd):
    """Several kernels carry the time point in gridDim.y (at most 65535 on this runtime): a grid of 70 000 steps -- the
    reference runs such grids at low order (examples/cnot3_optimize_gate.sb:27-40) -- is split into windows by itself even
    though it fits the memory, and equals the same grid under an explicit budget (more, shorter windows)."""
    prob, ctrl, pcof, target = cases.cnot2_case(qgd, nsteps=70000, tf=100.0)
    dp = qgd.DeviceProblem(prob, 2); dp.set_controls(ctrl); dp.set_target(target)
    plan = dp.memory_plan()
    assert plan["windows"] >= 2 and plan["steps_per_window"] <= 65000, plan
    g, o = dp.discrete_adjoint(pcof)
    f = dp.eval_forward(pcof)
    assert np.abs(np.asarray(f) - np.asarray(o)).max() <= 1e-12 * max(1.0, np.abs(np.asarray(o)).max())
    dp.close()
    dp = qgd.DeviceProblem(prob, 2)
    dp.set_memory_budget(plan["window_bytes"] // 3)
    dp.set_controls(ctrl); dp.set_target(target)
    assert dp.memory_plan()["windows"] > plan["windows"]
    g2, o2 = dp.discrete_adjoint(pcof)
    dp.close()
    assert np.abs(g2 - g).max() <= 1e-11 * np.abs(g).max()
    assert np.abs(np.asarray(o2) - np.asarray(o)).max() <= 1e-12 * max(1.0, np.abs(np.asarray(o)).max())


@pytest.mark.parametrize("which,order,nsteps,windows", [("cnot2", 6, 90, 3), ("guarded", 6, 64, 2)])
def test_control_tables_on_a_windowed_grid(qgd, which, order, nsteps, windows):
    """Controls that are not linear in their coefficients reach the library as tables + Jacobian per evaluation
    (qgd_set_control_tables + qgd_set_control_basis, NULL pcof: Control.jl:6-27 leaves the family open).  On a windowed grid the
    caller's tables cover the WHOLE grid and every window uploads its slice: gradient, scalars, forward-only evaluation and
    eval_adjoint equal the resident handle's -- for a linear family seen through its pointwise protocol only (which must also
    reproduce the basis path) and for a sine control."""
    prob, ctrl, pcof, target = getattr(cases, which + "_case")(qgd, nsteps=nsteps, tf=float(nsteps) / (1 if which.startswith("cnot") else 2))
    for kind in ("pointwise", "sine"):
        if kind == "pointwise":
            gctrl, gp = [cases.PointwiseOnly(c) for c in ctrl], pcof
        else:
            gctrl = [cases.SineControl(prob.tf) for _ in ctrl]
            gp = np.concatenate([[0.02 * (k + 1), 0.3 + 0.1 * k, 0.2 * k] for k in range(len(ctrl))])
        ref = qgd.DeviceProblem(prob, order); ref.set_controls(gctrl); ref.set_target(target)
        g_ref, o_ref = ref.discrete_adjoint(gp)
        f_ref = np.asarray(ref.eval_forward(gp))
        full = ref.memory_plan()["window_bytes"]
        dp = qgd.DeviceProblem(prob, order); dp.set_memory_budget(int(full / windows * 1.15)); dp.set_controls(gctrl); dp.set_target(target)
        assert dp.memory_plan()["windows"] >= 2
        for rep in range(2):
            g, o = dp.discrete_adjoint(gp)
            assert np.abs(g - g_ref).max() <= 1e-11 * np.abs(g_ref).max(), (kind, rep)
            assert np.abs(np.asarray(o) - np.asarray(o_ref)).max() <= 1e-12 * max(1.0, np.abs(np.asarray(o_ref)).max())
        assert np.abs(np.asarray(dp.eval_forward(gp)) - f_ref).max() <= 1e-12 * max(1.0, np.abs(f_ref).max())
        term = np.random.default_rng(3).standard_normal((prob.real_system_size, prob.N_initial_conditions))
        l_ref, l_win = ref.eval_adjoint(gp, term), dp.eval_adjoint(gp, term)
        assert np.abs(l_win[:, 0] - l_ref[:, 0]).max() <= 1e-11 * max(1.0, np.abs(l_ref[:, 0]).max())
        if kind == "pointwise":      # the general path reproduces the basis path
            lin = qgd.DeviceProblem(prob, order); lin.set_controls(ctrl); lin.set_target(target)
            g_lin, _ = lin.discrete_adjoint(pcof)
            lin.close()
            assert np.abs(g_ref - g_lin).max() <= 1e-11 * np.abs(g_lin).max()
        ref.close(); dp.close()
