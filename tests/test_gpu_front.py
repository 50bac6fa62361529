"""GPU tests of the fused front (csrc/qgd_front.h, round 6): N = 64 problems with sparse operators evaluate with the same-point
step propagator S_n = R_n L_n^-1 -- one workgroup per time point builds L_n^H, R_n^H and eliminates [L_n^H | R_n^H]; the forward
sweep runs in phi = L psi, k_psi turns it into the state history, guard forcing and h = L^-H f; the adjoint sweep runs in lambda.
Reference being reproduced: src/forward_evolution.jl:181-220 (forward step), :421-461 (adjoint step), checked here against the
general path of the library (two-point propagators, QGD_PATHS=no_front), the numpy statements of tests/proto_propagator.py and the
CPU oracle.  The library takes the front by itself on grids of 513 .. 704 time points (where it wins: qgd_host_eval.cpp,
front_applies); QGD_PATHS=front takes it wherever it is supported, which is how the short grids here reach it."""
import os

import numpy as np
import pytest

import cases
import proto_propagator as pp

pytestmark = pytest.mark.gpu


def _evaluate(qgd, prob, ctrl, pcof, target, order, paths, monkeypatch, outputs=True, calls=1):
    """One handle under QGD_PATHS=paths: gradient, scalars, (uv_history, lambda_history, adjoint_forcing), which path ran, repivoted."""
    monkeypatch.setenv("QGD_PATHS", paths)
    dp = qgd.DeviceProblem(prob, order)
    dp.set_target(target); dp.set_controls(ctrl)
    out = {}
    if outputs:
        hist = np.zeros(dp._hist_shape(), order="F"); lam = np.zeros(dp._hist_shape(), order="F")
        forc = np.zeros((2 * dp.N, prob.nsteps + 1, dp.c), order="F")
        g, o3 = dp.discrete_adjoint(pcof, uv_history=hist, lambda_history=lam, adjoint_forcing=forc)
        out.update(hist=hist, lam=lam, forc=forc)
    for _ in range(calls):
        g, o3 = dp.discrete_adjoint(pcof)
    out.update(g=g, o3=np.asarray(o3), front=dp.front_path_taken(), rep=int(dp.intermediate("repivoted")))
    dp.close()
    monkeypatch.delenv("QGD_PATHS")
    return out


@pytest.mark.parametrize("nsteps,order", [(6, 8), (40, 8), (300, 8), (550, 8), (800, 8), (40, 2), (40, 4), (40, 6)])
def test_front_equals_general_path(qgd, orc, monkeypatch, nsteps, order):
    """cnot3 (the headline problem) on grids that run every shape of the launch -- fewer workgroups than CUs, one round of two,
    of two to three (550: the tail workgroups start from step matrices built by the tables launch), more than one round (800) --
    and at every Hermite order the front is compiled for: gradient, objective scalars, state history with stage derivatives,
    lambda and guard forcing equal the general path's to rounding; on the short grids also the numpy statement of the local
    form and the oracle (1e-10)."""
    prob, ctrl, pcof, target = cases.cnot3_case(qgd, nsteps=nsteps, tf=float(nsteps))
    f = _evaluate(qgd, prob, ctrl, pcof, target, order, "front", monkeypatch)
    g = _evaluate(qgd, prob, ctrl, pcof, target, order, "no_front", monkeypatch)
    assert f["front"] and not g["front"] and f["rep"] == 0
    gs = np.abs(g["g"]).max()
    assert np.abs(f["g"] - g["g"]).max() <= 1e-12 * gs
    assert np.abs(f["o3"] - g["o3"]).max() <= 1e-12
    assert np.abs(f["hist"] - g["hist"]).max() <= 1e-12 * max(1.0, np.abs(g["hist"]).max())
    assert np.abs(f["lam"] - g["lam"]).max() <= 1e-12 * np.abs(g["lam"]).max()
    assert np.abs(f["forc"] - g["forc"]).max() <= 1e-14
    if nsteps <= 40:
        Gp, Gq, off = qgd.control_basis(ctrl, prob.nsteps, prob.tf, order // 2)
        r = pp.evaluate_local(prob, Gp, Gq, off, pcof, target, order)
        assert np.abs(f["g"] - r["grad"]).max() <= 1e-11 * gs
        assert np.abs(pp.history_real(r["ws"]) - f["hist"]).max() <= 1e-11 * max(1.0, np.abs(f["hist"]).max())
        assert abs(f["o3"][2] - r["guard"]) <= 1e-13 and abs(1 - (f["o3"][0] ** 2 + f["o3"][1] ** 2) / prob.N_ess_levels ** 2 - r["infidelity"]) <= 1e-12
        if nsteps <= 6 or order <= 4:
            assert cases.oracle_pins(orc, prob, ctrl, pcof, target, order, r)


def test_front_gradient_is_bitwise_reproducible(qgd, monkeypatch):
    """Two handles, several evaluations each: the same bits (no atomics between the kernels and the gradient: the guard
    partials of k_psi and the rows of the basis contraction are added in a fixed order)."""
    prob, ctrl, pcof, target = cases.cnot3_case(qgd, nsteps=550, tf=550.0)
    a = _evaluate(qgd, prob, ctrl, pcof, target, 8, "", monkeypatch, outputs=False, calls=3)
    b = _evaluate(qgd, prob, ctrl, pcof, target, 8, "", monkeypatch, outputs=False, calls=2)
    assert a["front"] and b["front"]
    assert np.array_equal(a["g"], b["g"]) and np.array_equal(a["o3"], b["o3"])


def _paired_drift(qgd, partner, noise, nsteps=12, order=6):
    """cnot3's operators with a drift that pairs the levels i <-> i^partner, dt*h at the zero of Re q_3(iy) (the diagonal of
    L vanishes beside off-diagonal entries of modulus ~1): still a handful of entries per row, so the sparse kernels and the
    front take it.  noise * (the physical drift's diagonal) keeps the diagonal pivots merely tiny; noise = 0: exactly zero."""
    prob, ctrl, pcof, target = cases.cnot3_case(qgd, nsteps=nsteps, tf=0.8)
    dt = 0.8 / nsteps
    H = np.zeros((64, 64))
    for i in range(64):
        H[i, i ^ partner] = np.sqrt(10.0) / dt
    prob.system_sym = np.asfortranarray(H + noise * np.diag(np.diag(prob.system_sym)))
    return prob, ctrl, 0.1 * pcof, target, order


@pytest.mark.parametrize("partner,noise,stage", [(1, 0.1, 1), (1, 0.0, 1), (32, 0.1, 2), (32, 0.0, 2)])
def test_front_pivot_stages(qgd, orc, monkeypatch, partner, noise, stage):
    """The three pivot stages of the column-block elimination inside the front kernel, and the re-layout of the propagator
    behind the two pivoted ones (qgd_k_inverse.hip: front_relayout).  Pairing i <-> i^1: the large entries lie inside the
    16 x 16 diagonal tiles, partial pivoting in the tile does every matrix; i <-> i^32: outside, the fully pivoted elimination
    does.  noise = 0 is the advisor's case (round 5): the diagonal pivots are EXACTLY zero, the diagonal attempt must notice
    (1 / 0 gives NaN multipliers, which the growth check alone does not see) instead of passing NaNs on."""
    prob, ctrl, pcof, target, order = _paired_drift(qgd, partner, noise)
    Gp, Gq, off = qgd.control_basis(ctrl, prob.nsteps, prob.tf, order // 2)
    ref = pp.evaluate_local(prob, Gp, Gq, off, pcof, target, order)
    if partner == 1 and noise > 0:
        assert cases.oracle_pins(orc, prob, ctrl, pcof, target, order, ref)
    f = _evaluate(qgd, prob, ctrl, pcof, target, order, "front", monkeypatch, calls=2)      # (the second call starts with pivoting: same result)
    g = _evaluate(qgd, prob, ctrl, pcof, target, order, "no_front", monkeypatch)
    nt = prob.nsteps + 1
    assert f["front"] and not g["front"]
    assert (f["rep"] & 0xFFFF) == nt and (f["rep"] >> 16) == (nt if stage == 2 else 0), (partner, noise, f["rep"])
    gs = np.abs(ref["grad"]).max()
    assert np.isfinite(f["g"]).all()
    assert np.abs(f["g"] - ref["grad"]).max() <= 1e-10 * gs, (partner, noise)
    assert np.abs(f["g"] - g["g"]).max() <= 1e-10 * gs
    assert np.abs(f["hist"] - pp.history_real(ref["ws"])).max() <= 1e-10 * max(1.0, np.abs(f["hist"]).max())


def test_front_without_guard_many_columns(qgd, orc, monkeypatch):
    """All 64 levels essential: no guard projector (the adjoint sweep's forcing stays zero, k_psi does the one product), 64
    initial conditions = 8 column groups (the terminal value L_N^-H target group by group, the overlaps summed over all)."""
    freqs = 2 * np.pi * np.array([4.10595, 4.81526, 7.8447])
    kerr = 2 * np.pi * np.array([[0.2198, 1e-6, 0.0025], [1e-6, 0.2252, 0.0025], [0.0025, 0.0025, 3e-5]])
    nsteps = 30
    prob = qgd.DispersiveProblem((4, 4, 4), (4, 4, 4), freqs, freqs, kerr, float(nsteps), nsteps, sparse_rep=False,
                                 gmres_abstol=1e-15, gmres_reltol=1e-15)
    assert prob.N_initial_conditions == 64 and not np.any(prob.guard_subspace_projector)
    ctrl = cases.cnot3_controls(qgd, prob)
    rng = np.random.default_rng(3)
    pcof = (rng.random(qgd.get_number_of_control_parameters(ctrl)) - 0.5) * 2 * np.pi * 0.005
    target = rng.random((64, 64)) + 1j * rng.random((64, 64))
    f = _evaluate(qgd, prob, ctrl, pcof, target, 8, "front", monkeypatch)
    g = _evaluate(qgd, prob, ctrl, pcof, target, 8, "no_front", monkeypatch)
    assert f["front"] and not g["front"]
    gs = np.abs(g["g"]).max()
    assert np.abs(f["g"] - g["g"]).max() <= 1e-12 * gs and np.abs(f["o3"] - g["o3"]).max() <= 1e-11
    assert np.abs(f["lam"] - g["lam"]).max() <= 1e-12 * np.abs(g["lam"]).max() and not np.any(f["forc"])
    Gp, Gq, off = qgd.control_basis(ctrl, prob.nsteps, prob.tf, 4)
    r = pp.evaluate_local(prob, Gp, Gq, off, pcof, target, 8)
    assert np.abs(f["g"] - r["grad"]).max() <= 1e-11 * gs


def test_front_eligibility_and_reuse(qgd, monkeypatch):
    """Which calls take the front, and what the other entry points see afterwards.  (1) qgd_eval_forward and
    qgd_discrete_adjoint do; history_precomputed with the same pcof reuses the front's forward sweep (terminal value formed by
    the stand-alone terminal kernel).  (2) The 4-pivot panel inverse (QGD_PATHS=inv_panels), a windowed grid: the general path.  (3) qgd_get_intermediate("P" / "Linv" / "L") after a front evaluation returns the TWO-POINT form's
    matrices (the forward evaluation is redone on the general path), and the next evaluation takes the front again.
    (4) The forced gradient and eval_adjoint (general path) agree with the front's gradient / lambda."""
    # the library's own choice: the front on 513 .. 704 time points, the general path on shorter and longer grids
    for nsteps, want in ((300, False), (520, True), (800, False)):
        p_, c_, x_, t_ = cases.cnot3_case(qgd, nsteps=nsteps, tf=float(nsteps))
        d_ = qgd.DeviceProblem(p_, 8); d_.set_target(t_); d_.set_controls(c_)
        d_.discrete_adjoint(x_)
        assert d_.front_path_taken() == want, nsteps
        d_.close()
    monkeypatch.setenv("QGD_PATHS", "front")
    prob, ctrl, pcof, target = cases.cnot3_case(qgd, nsteps=24, tf=24.0)
    order = 8
    Gp, Gq, off = qgd.control_basis(ctrl, prob.nsteps, prob.tf, order // 2)
    ref = pp.evaluate(prob, Gp, Gq, off, pcof, target, order)
    gs = np.abs(ref["grad"]).max()
    dp = qgd.DeviceProblem(prob, order); dp.set_target(target); dp.set_controls(ctrl)
    o3 = dp.eval_forward(pcof)
    assert dp.front_path_taken()
    g, o3b = dp.discrete_adjoint(pcof, history_precomputed=True)
    assert dp.front_path_taken() and np.abs(g - ref["grad"]).max() <= 1e-11 * gs and np.abs(np.asarray(o3) - np.asarray(o3b)).max() <= 1e-13
    P = dp.intermediate("P"); Linv = dp.intermediate("Linv"); L = dp.intermediate("L")
    assert not dp.front_path_taken()
    assert np.abs(P[:prob.nsteps] - ref["P"]).max() <= 1e-11 and np.abs(Linv[1:] - ref["Linv"][1:]).max() <= 1e-11 and np.abs(L - ref["L"]).max() <= 1e-12
    g2, _ = dp.discrete_adjoint(pcof)
    assert dp.front_path_taken() and np.abs(g2 - g).max() <= 1e-13 * gs
    gf = dp.eval_grad_forced(pcof)
    assert not dp.front_path_taken() and np.abs(gf - g).max() <= 1e-11 * gs
    g3, _ = dp.discrete_adjoint(pcof)
    assert dp.front_path_taken() and np.array_equal(g3, g2)
    # a stored front sweep is not reused across a change of cost type or target (k_psi formed its terminal value for the cost
    # type and the target it was computed with): history_precomputed then redoes the sweep
    dp.set_cost_type("Tracking")
    gt, _ = dp.discrete_adjoint(pcof, history_precomputed=True)
    assert dp.front_path_taken()
    gt2, _ = dp.discrete_adjoint(pcof)
    assert np.array_equal(gt, gt2) and np.abs(gt - g).max() > 1e-3 * gs
    dp.set_cost_type("Infidelity")
    dp.eval_forward(pcof)
    t2 = target[:, ::-1].copy()
    dp.set_target(t2)
    gn, _ = dp.discrete_adjoint(pcof, history_precomputed=True)
    refn = pp.evaluate(prob, Gp, Gq, off, pcof, t2, order)
    assert np.abs(gn - refn["grad"]).max() <= 1e-11 * np.abs(refn["grad"]).max()
    dp.set_target(target)
    dp.set_memory_budget(40 << 20)
    if dp.memory_plan()["windows"] > 1:
        g4, _ = dp.discrete_adjoint(pcof)
        assert not dp.front_path_taken() and np.abs(g4 - g).max() <= 1e-11 * gs
    dp.close()
    monkeypatch.setenv("QGD_PATHS", "inv_panels")
    dp = qgd.DeviceProblem(prob, order); dp.set_target(target); dp.set_controls(ctrl)
    g5, _ = dp.discrete_adjoint(pcof)
    assert not dp.front_path_taken() and np.abs(g5 - g).max() <= 1e-12 * gs
    dp.close()


def test_front_non_finite_coefficients(qgd, monkeypatch):
    """A coefficient vector with a NaN / an infinity in it: the evaluation ENDS with non-finite results, and the next
    evaluation of the same handle is bit for bit what it was before."""
    prob, ctrl, pcof, target = cases.cnot3_case(qgd, nsteps=40, tf=40.0)
    monkeypatch.setenv("QGD_PATHS", "front")
    dp = qgd.DeviceProblem(prob, 8); dp.set_controls(ctrl); dp.set_target(target)
    g0, o0 = dp.discrete_adjoint(pcof)
    assert dp.front_path_taken()
    for bad in (np.nan, np.inf, 1e200):
        p2 = pcof.copy(); p2[3] = bad
        g, o = dp.discrete_adjoint(p2)
        assert not np.isfinite(g).all() and not np.isfinite(o[0])
        g1, o1 = dp.discrete_adjoint(pcof)
        assert np.array_equal(g1, g0) and np.array_equal(np.asarray(o1), np.asarray(o0)), bad
    dp.close()


def test_front_padded_problem(qgd, orc, monkeypatch):
    """N = 60 = (5, 4, 3) levels in 64 x 64 tiles (rows and columns 60..63 are padding: identity in L and R, zero in the states),
    two of three controls' worth of guard levels, order 6: the front against the general path and the oracle."""
    freqs = 2 * np.pi * np.array([4.10595, 4.81526, 7.8447])
    kerr = 2 * np.pi * np.array([[0.2198, 1e-6, 0.0025], [1e-6, 0.2252, 0.0025], [0.0025, 0.0025, 3e-5]])
    nsteps = 10
    prob = qgd.DispersiveProblem((5, 4, 3), (2, 2, 2), freqs, freqs, kerr, float(nsteps), nsteps, sparse_rep=False,
                                 gmres_abstol=1e-15, gmres_reltol=1e-15)
    assert prob.N_tot_levels == 60
    ctrl = cases.cnot3_controls(qgd, prob)
    rng = np.random.default_rng(5)
    pcof = (rng.random(qgd.get_number_of_control_parameters(ctrl)) - 0.5) * 2 * np.pi * 0.005
    target = cases.rand_target(prob, 7)
    order = 6
    f = _evaluate(qgd, prob, ctrl, pcof, target, order, "front", monkeypatch)
    g = _evaluate(qgd, prob, ctrl, pcof, target, order, "no_front", monkeypatch)
    assert f["front"] and not g["front"]
    gs = np.abs(g["g"]).max()
    assert np.abs(f["g"] - g["g"]).max() <= 1e-12 * gs and np.abs(f["o3"] - g["o3"]).max() <= 1e-12
    assert np.abs(f["hist"] - g["hist"]).max() <= 1e-12 and np.abs(f["lam"] - g["lam"]).max() <= 1e-12 * np.abs(g["lam"]).max()
    Gp, Gq, off = qgd.control_basis(ctrl, prob.nsteps, prob.tf, order // 2)
    r = pp.evaluate_local(prob, Gp, Gq, off, pcof, target, order)
    assert np.abs(f["g"] - r["grad"]).max() <= 1e-11 * gs
    assert cases.oracle_pins(orc, prob, ctrl, pcof, target, order, r)


@pytest.mark.parametrize("cost_type", ["Tracking", "Norm"])
def test_front_cost_types(qgd, orc, monkeypatch, cost_type):
    """:Tracking and :Norm (eval_grad_discrete_adjoint.jl:26-35: a terminal right-hand side that is local in the columns,
    -(psi_N - target) or -psi_N): k_psi forms L_N^-H rhs for the terminal value.  Front against the general path, and the
    general path's pin -- the oracle -- on the same inputs."""
    prob, ctrl, pcof, target = cases.cnot3_case(qgd, nsteps=16, tf=16.0)
    res = {}
    for tag, paths in (("front", "front"), ("general", "no_front")):
        monkeypatch.setenv("QGD_PATHS", paths)
        dp = qgd.DeviceProblem(prob, 8); dp.set_target(target); dp.set_controls(ctrl); dp.set_cost_type(cost_type)
        lam = np.zeros(dp._hist_shape(), order="F")
        g, o3 = dp.discrete_adjoint(pcof, lambda_history=lam)
        res[tag] = (g, np.asarray(o3), lam, dp.front_path_taken())
        dp.close()
        monkeypatch.delenv("QGD_PATHS")
    f, g = res["front"], res["general"]
    assert f[3] and not g[3]
    assert np.abs(f[0] - g[0]).max() <= 1e-12 * np.abs(g[0]).max() and np.abs(f[1] - g[1]).max() <= 1e-12
    assert np.abs(f[2] - g[2]).max() <= 1e-12 * np.abs(g[2]).max()
    orc.set_converged_terminal(True); orc.set_num_threads(8); orc.set_cost_type(cost_type)
    try:
        go = orc.discrete_adjoint(prob, ctrl, pcof, target, order=8)
    finally:
        orc.set_converged_terminal(False); orc.set_num_threads(0); orc.set_cost_type("Infidelity")
    assert np.abs(f[0] - go).max() <= 1e-10 * np.abs(go).max()
