"""GPU tests of the small-problem path (csrc/qgd_k_tiny.hip, qgd_set_small_path): N <= 4 levels, <= 4 initial conditions,
<= 128 time points -- the reference's Rabi oscillator and its two-qubit CNOT (examples/cnot2_optimization.jl,
BASELINE.json configs[0..1]) -- evaluated in four launches on the vector ALU, one thread per (time point, column).  Compared with the ORACLE (the reference's
algorithm restated on the CPU) and with the general device path (the same problems are what tests/test_gpu_parity.py runs
on the general kernels: conftest.py switches this path off for the rest of the suite)."""
import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu


def _handles(qgd, prob, order, ctrl, target, cost_type="Infidelity"):
    out = []
    for small in (False, True):
        dp = qgd.DeviceProblem(prob, order)
        dp.set_small_path(small)
        dp.set_controls(ctrl)
        if target is not None:
            dp.set_target(target)
        dp.set_cost_type(cost_type)
        out.append(dp)
    return out


def _selected(dp):
    return bool(dp.small_path_taken())


@pytest.mark.parametrize("order,nsteps", [(2, 100), (4, 100), (8, 100), (10, 100), (12, 100)])
def test_cnot2_full_grid_vs_oracle_and_general_path(qgd, orc, order, nsteps):
    """BASELINE.json configs[0..1]: cnot2 (N = 4, 4 columns, 2 controls x 22 coefficients, dt = 1, 100 steps) at orders
    2-12: gradient against the
    oracle's discrete adjoint 1e-10, scalars 1e-12, and against the general device path 1e-12; forward-only evaluation; the
    same bits on every run."""
    prob, ctrl, pcof, target = cases.cnot2_case(qgd, nsteps=nsteps, tf=float(nsteps), amp=1e-2)
    orc.set_converged_terminal(True)
    try:
        g_ref, h_ref, _, _, _ = orc.discrete_adjoint(prob, ctrl, pcof, target, order=order, return_all=True)
    finally:
        orc.set_converged_terminal(False)
    gen, tiny = _handles(qgd, prob, order, ctrl, target)
    g0, o0 = gen.discrete_adjoint(pcof)
    assert not _selected(gen)
    g1, o1 = tiny.discrete_adjoint(pcof)
    assert _selected(tiny)
    assert np.abs(g1 - g_ref).max() <= 1e-10 * np.abs(g_ref).max()
    assert np.abs(g1 - g0).max() <= 1e-12 * np.abs(g0).max()
    assert np.abs(np.asarray(o1) - np.asarray(o0)).max() <= 1e-12
    infid = 1 - (o1[0] ** 2 + o1[1] ** 2) / prob.N_ess_levels ** 2
    assert abs(infid - orc.infidelity_real(h_ref[:, 0, -1, :], orc.target_real(target), prob.N_ess_levels)) <= 1e-12
    f1 = tiny.eval_forward(pcof)
    assert np.abs(np.asarray(f1) - np.asarray(o1)).max() <= 1e-14
    for _ in range(3):
        g2, o2 = tiny.discrete_adjoint(pcof)
        assert np.array_equal(g2, g1) and np.array_equal(np.asarray(o2), np.asarray(o1))
    g3, _ = tiny.discrete_adjoint(0.5 * pcof)
    g3_ref, _ = gen.discrete_adjoint(0.5 * pcof)
    assert np.abs(g3 - g3_ref).max() <= 1e-12 * np.abs(g3_ref).max()
    gen.close(); tiny.close()


def test_reference_gradient_cases(qgd, orc):
    """The problems of the reference's own gradient tests (test/GradientTests/compare_gradients.jl:103-230: Rabi and random
    N = 4 with GRAPE, degree-16 B-spline and carrier controls, 10 steps) at orders 2-12 on the small-problem path: the
    reference's contract -- adjoint against the oracle's adjoint and forced gradients."""
    for name, prob, ctrl, pcof, target in cases.gradient_cases(qgd):
        for order in (2, 4, 6, 8, 10, 12):
            g_ref = orc.discrete_adjoint(prob, ctrl, pcof, target, order=order)
            gf_ref = orc.eval_grad_forced(prob, ctrl, pcof, target, order=order)
            dp = qgd.DeviceProblem(prob, order); dp.set_small_path(True); dp.set_controls(ctrl); dp.set_target(target)
            g, _ = dp.discrete_adjoint(pcof)
            taken = _selected(dp)
            dp.close()
            n_pcof = qgd.get_number_of_control_parameters(ctrl)
            assert taken == (n_pcof <= 64), (name, order, n_pcof)      # (the carrier controls have 200 coefficients: general path)
            scale = np.abs(g_ref).max()
            assert np.abs(g - g_ref).max() <= 1e-10 * scale, (name, order)
            assert np.abs(g - gf_ref).max() <= 1e-10 * scale, (name, order)


@pytest.mark.parametrize("N,c,n_ops,order,nsteps,cost", [(4, 4, 2, 6, 40, "Infidelity"), (3, 2, 1, 4, 25, "Tracking"), (4, 3, 3, 8, 60, "Norm"),
                                                         (2, 2, 4, 2, 100, "Infidelity"), (4, 1, 2, 10, 17, "Infidelity"), (3, 3, 2, 12, 31, "Tracking")])
def test_shapes_guards_and_cost_types(qgd, orc, N, c, n_ops, order, nsteps, cost):
    """Random dense problems of every shape the path covers, with a diagonal guard projector (penalty, forcing and the
    affine part of the adjoint scan), fewer columns than rows, 1-4 control operators, the three cost types: against the
    oracle (1e-10) and the general path (1e-12)."""
    prob = qgd.construct_rand_prob(N, n_ops, tf=1.0, nsteps=nsteps, gmres_abstol=1e-15, gmres_reltol=1e-15, scale=0.7)
    prob.u0 = np.asfortranarray(prob.u0[:, :c]); prob.v0 = np.asfortranarray(prob.v0[:, :c]); prob.N_initial_conditions = c
    rng = np.random.default_rng(N * 100 + c)
    W = np.zeros((2 * N, 2 * N))
    wd = rng.random(N) * (rng.random(N) > 0.4)
    W[np.arange(N), np.arange(N)] = wd; W[N + np.arange(N), N + np.arange(N)] = wd
    prob.guard_subspace_projector = W
    ctrl = [qgd.FortranBSplineControl(3, 5, prob.tf) for _ in range(n_ops)]
    pcof = rng.standard_normal(qgd.get_number_of_control_parameters(ctrl))
    target = cases.rand_target(prob)
    orc.set_converged_terminal(True); orc.set_cost_type(cost)
    try:
        g_ref = orc.discrete_adjoint(prob, ctrl, pcof, target, order=order)
    finally:
        orc.set_converged_terminal(False); orc.set_cost_type("Infidelity")
    gen, tiny = _handles(qgd, prob, order, ctrl, target, cost)
    g0, o0 = gen.discrete_adjoint(pcof)
    g1, o1 = tiny.discrete_adjoint(pcof)
    assert _selected(tiny) and not _selected(gen)
    assert np.abs(g1 - g_ref).max() <= 1e-10 * np.abs(g_ref).max()
    assert np.abs(g1 - g0).max() <= 1e-12 * np.abs(g0).max()
    assert np.abs(np.asarray(o1) - np.asarray(o0)).max() <= 1e-12 * max(1.0, np.abs(np.asarray(o0)).max())
    assert o1[2] > 0 or not wd.any()
    gen.close(); tiny.close()


def test_what_the_single_kernel_path_leaves_to_the_general_one(qgd):
    """Output arrays, history_precomputed, diagnostics, event bracketing and problems outside its limits behave as before:
    the reference-shaped call after a small-problem evaluation redoes the sweep on the general path (and is right), the
    intermediates are those of the same evaluation, a grid too long for one workgroup is not taken."""
    prob, ctrl, pcof, target = cases.cnot2_case(qgd, nsteps=60, tf=60.0)
    order = 8
    gen, tiny = _handles(qgd, prob, order, ctrl, target)
    g0, o0 = gen.discrete_adjoint(pcof)
    shape = (prob.real_system_size, 1 + order // 2, 1 + prob.nsteps, prob.N_initial_conditions)
    ref = [np.zeros(shape, order="F"), np.zeros(shape, order="F"), np.zeros((shape[0], shape[2], shape[3]), order="F")]
    gen.discrete_adjoint(pcof, False, *ref)
    f = tiny.eval_forward(pcof)                          # single kernel
    assert _selected(tiny)
    got = [np.full(shape, np.nan, order="F"), np.full(shape, np.nan, order="F"), np.full((shape[0], shape[2], shape[3]), np.nan, order="F")]
    g1, o1 = tiny.discrete_adjoint(pcof, True, *got)     # history_precomputed + outputs: general path, sweep redone
    assert not _selected(tiny)
    assert np.abs(g1 - g0).max() <= 1e-12 * np.abs(g0).max()
    for a, b in zip(got, ref):
        assert np.isfinite(a).all() and np.abs(a - b).max() <= 1e-13 * max(1.0, np.abs(b).max())
    g2, _ = tiny.discrete_adjoint(pcof)                  # single kernel again ...
    assert _selected(tiny)
    P_t, P_g = tiny.intermediate("P"), gen.intermediate("P")      # ... whose intermediates come from a general-path rerun
    assert np.abs(P_t - P_g).max() <= 1e-13
    tiny.set_timing(1)
    g3, _ = tiny.discrete_adjoint(pcof)                  # bracketing on: general path (it has the phases)
    assert not _selected(tiny) and "inverse" in tiny.timings()
    tiny.set_timing(0)
    assert np.abs(g3 - g0).max() <= 1e-12 * np.abs(g0).max() and np.abs(g2 - g0).max() <= 1e-12 * np.abs(g0).max()
    gen.close(); tiny.close()
    prob, ctrl, pcof, target = cases.cnot2_case(qgd, nsteps=400, tf=100.0)      # 401 time points: more than the scan workgroup holds
    gen, tiny = _handles(qgd, prob, order, ctrl, target)
    ga, _ = gen.discrete_adjoint(pcof); gb, _ = tiny.discrete_adjoint(pcof)
    assert not _selected(tiny) and np.array_equal(ga, gb)
    gen.close(); tiny.close()


@pytest.mark.parametrize("nsteps,taken", [(1, False), (2, True), (3, True), (4, True), (7, True), (127, True), (128, False)])
def test_grid_limits(qgd, orc, nsteps, taken):
    """The shortest grids the scan takes (3 time points: one step on either side of the middle one), odd lengths that leave
    its last block short, the longest one (128 time points = 512 threads for 4 columns), and the lengths on either side that
    go to the general path: against the general path (1e-12) and, below 10 steps, against the oracle (1e-10); after
    qgd_set_nsteps on the same handle too."""
    prob, ctrl, pcof, target = cases.cnot2_case(qgd, nsteps=nsteps, tf=0.5 * nsteps, amp=2e-2)
    order = 6
    gen, tiny = _handles(qgd, prob, order, ctrl, target)
    g0, o0 = gen.discrete_adjoint(pcof)
    g1, o1 = tiny.discrete_adjoint(pcof)
    assert _selected(tiny) == taken, nsteps
    assert np.abs(g1 - g0).max() <= 1e-12 * np.abs(g0).max()
    assert np.abs(np.asarray(o1) - np.asarray(o0)).max() <= 1e-12
    f1 = tiny.eval_forward(pcof)
    assert np.abs(np.asarray(f1) - np.asarray(o0)).max() <= 1e-12
    if nsteps < 10:
        orc.set_converged_terminal(True)
        try:
            g_ref = orc.discrete_adjoint(prob, ctrl, pcof, target, order=order)
        finally:
            orc.set_converged_terminal(False)
        assert np.abs(g1 - g_ref).max() <= 1e-10 * np.abs(g_ref).max()
    gen.close(); tiny.close()


def test_nan_coefficients_propagate_like_the_general_path(qgd):
    """A NaN control coefficient (an optimizer probing outside its trust region) gives NaN results and NO error on both paths --
    what the reference's GMRES does with it -- and leaves the handle usable."""
    prob, ctrl, pcof, target = cases.cnot2_case(qgd, nsteps=40, tf=40.0)
    bad = pcof.copy(); bad[3] = np.nan
    for dp in _handles(qgd, prob, 8, ctrl, target):
        g, o = dp.discrete_adjoint(bad)
        assert not np.isfinite(g).all() and not np.isfinite(o[0])
        g, o = dp.discrete_adjoint(pcof)
        assert np.isfinite(g).all() and np.isfinite(np.asarray(o)).all()
        dp.close()


def test_results_by_copy_equal_results_by_mirror(qgd, monkeypatch):
    """QGD_RESULT_MIRROR=0 (read when the handle is created: results by a device-to-host copy + stream synchronisation instead of
    the host-memory mirror the last kernel writes): the same bits, gradient evaluations and forward-only ones, changing pcof."""
    prob, ctrl, pcof, target = cases.cnot2_case(qgd, nsteps=60, tf=60.0)
    res = {}
    for label in ("mirror", "copy"):
        if label == "copy":
            monkeypatch.setenv("QGD_RESULT_MIRROR", "0")
        dp = qgd.DeviceProblem(prob, 8)
        monkeypatch.delenv("QGD_RESULT_MIRROR", raising=False)
        dp.set_small_path(True); dp.set_controls(ctrl); dp.set_target(target)
        out = []
        for s in (1.0, 0.5, 1.0):
            g, o = dp.discrete_adjoint(s * pcof)
            assert _selected(dp)
            out += [g, np.asarray(o), np.asarray(dp.eval_forward(s * pcof))]
        res[label] = out
        dp.close()
    assert all(np.array_equal(a, b) for a, b in zip(res["mirror"], res["copy"]))
