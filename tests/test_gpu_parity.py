"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle
on the same seeded inputs.  Tolerances (fp64, stated by BASELINE.json's
north_star): gradient within 1e-10 relative; histories within 1e-11 absolute
(the oracle's own GMRES stops at 1e-15 absolute residual)."""
import numpy as np
import pytest

import cases
import proto_propagator as pp

pytestmark = pytest.mark.gpu

GRAD_RTOL = 1e-10
HIST_ATOL = 1e-11


def close(a, ref, tol=HIST_ATOL):
    """max |a - ref| <= tol * max(1, max|ref|): absolute for O(1) states, relative for the
    large high-order Taylor coefficients of the random test problems."""
    return np.abs(a - ref).max() <= tol * max(1.0, np.abs(ref).max())


def _loaded(qgd):
    lib = qgd._lib.lib()
    assert lib.qgd_abi_version() == 2


def test_library_loaded(qgd):
    _loaded(qgd)


@pytest.mark.parametrize("which", ["cnot2", "guarded", "cnot3"])
def test_apply_hamiltonian(qgd, orc, which):
    """qgd_apply_hamiltonian vs the oracle's apply_hamiltonian! (hermite.jl:556-588), every
    derivative order, both signs, all columns at once."""
    prob, ctrl, pcof, target = getattr(cases, which + "_case")(qgd)
    order = 8
    m = order // 2
    dp = qgd.DeviceProblem(prob, order)
    dp.set_controls(ctrl)
    dp.eval_forward(pcof)  # puts the control tables on the device
    rng = np.random.default_rng(5)
    w = rng.standard_normal((prob.real_system_size, prob.N_initial_conditions))
    Gp, Gq, off = qgd.control_basis(ctrl, prob.nsteps, prob.tf, m)
    tp, tq = pp.tables(Gp, Gq, off, pcof, m)
    for n in (0, prob.nsteps // 2, prob.nsteps):
        for d in range(m):
            for adj in (False, True):
                out = dp.apply_hamiltonian(w, n, d, adj)
                ref = orc.apply_hamiltonian(prob, tp[n].copy(), tq[n].copy(), w, d, adj)
                assert np.abs(out - ref).max() <= 1e-13 * max(1.0, np.abs(ref).max()), (which, n, d, adj)
    dp.close()


@pytest.mark.parametrize("which,order", [("cnot2", 2), ("cnot2", 8), ("guarded", 6), ("cnot3", 2), ("cnot3", 4),
                                         ("cnot3", 6), ("cnot3", 8), ("cnot3", 10), ("cnot3", 12)])
def test_stage_matrices(qgd, orc, which, order):
    """L(t_n), R(t_n), L^-1, P_n on the device vs the numpy statement of the same algorithm."""
    prob, ctrl, pcof, target = getattr(cases, which + "_case")(qgd)
    m = order // 2
    dp = qgd.DeviceProblem(prob, order)
    dp.set_controls(ctrl)
    dp.eval_forward(pcof)
    Gp, Gq, off = qgd.control_basis(ctrl, prob.nsteps, prob.tf, m)
    ref = pp.evaluate(prob, Gp, Gq, off, pcof, target, order)
    cases.oracle_pins(orc, prob, ctrl, pcof, target, order, ref)
    L, R, Li, P = (dp.intermediate(k) for k in ("L", "R", "Linv", "P"))
    assert np.abs(L - ref["L"]).max() < 1e-13
    assert np.abs(R - ref["R"]).max() < 1e-13
    assert np.abs(Li[1:] - ref["Linv"][1:]).max() < 1e-12
    assert np.abs(P[:-1] - ref["P"]).max() < 1e-12
    dp.close()


@pytest.mark.parametrize("order", [2, 4, 6, 8, 10])
def test_gradient_reference_cases(qgd, orc, order):
    """The reference's own gradient test problems (compare_gradients.jl:103-230): device
    discrete adjoint vs oracle discrete adjoint, and device history vs oracle history."""
    for name, prob, ctrl, pcof, target in cases.gradient_cases(qgd):
        g_ref, h_ref, lam_ref, f_ref, _ = orc.discrete_adjoint(prob, ctrl, pcof, target, order=order, return_all=True)
        shape = h_ref.shape
        hist = np.zeros(shape, order="F"); lam = np.zeros(shape, order="F")
        forcing = np.zeros((shape[0], shape[2], shape[3]), order="F")
        grad = np.zeros_like(g_ref)
        qgd.discrete_adjoint_(grad, hist, lam, forcing, prob, ctrl, pcof, target, order=order)
        assert close(hist, h_ref), (name, order)
        assert close(lam[:, 0], lam_ref[:, 0]), (name, order)
        assert close(forcing, f_ref), (name, order)
        assert np.abs(grad - g_ref).max() <= GRAD_RTOL * np.abs(g_ref).max(), (name, order)
    qgd.clear_cache()


@pytest.mark.parametrize("which,order", [("cnot2", 2), ("cnot2", 8), ("guarded", 4), ("guarded", 8), ("guarded", 12),
                                         ("dense_guard", 6), ("cnot3", 8)])
def test_configs_vs_oracle(qgd, orc, which, order):
    """BASELINE.json configurations (reduced nsteps so the oracle finishes in seconds).

    The oracle's terminal solve is run to convergence here: with the reference's gmres!
    defaults (restart 20, 2N iterations, eval_grad_discrete_adjoint.jl:61-62) it stops
    2.7e-6 short on cnot3 at dt = 1 (see test_oracle.py::test_reference_terminal_solve_stalls),
    which would mask a 1e-10 comparison."""
    prob, ctrl, pcof, target = getattr(cases, which + "_case")(qgd)
    orc.set_converged_terminal(True)
    try:
        g_ref, h_ref, lam_ref, f_ref, _ = orc.discrete_adjoint(prob, ctrl, pcof, target, order=order, return_all=True)
    finally:
        orc.set_converged_terminal(False)
    dp = qgd.device_problem(prob, order)
    dp.set_controls(ctrl)
    dp.set_target(target)
    hist = np.zeros(h_ref.shape, order="F")
    out3 = dp.eval_forward(pcof, hist)
    assert close(hist, h_ref)
    infid = 1 - (out3[0] ** 2 + out3[1] ** 2) / prob.N_ess_levels ** 2
    assert abs(infid - orc.infidelity_real(h_ref[:, 0, -1, :], orc.target_real(target), prob.N_ess_levels)) < 1e-12
    assert abs(out3[2] - orc.guard_penalty_real(prob, h_ref)) < 1e-12
    # gradient with the stored forward sweep (history_precomputed, ipopt_optimal_control.jl:297-308)
    grad, _ = dp.discrete_adjoint(pcof, history_precomputed=True)
    assert np.abs(grad - g_ref).max() <= GRAD_RTOL * np.abs(g_ref).max()
    # and from scratch through the reference-shaped call
    grad2 = qgd.discrete_adjoint(prob, ctrl, pcof, target, order=order)
    assert np.abs(grad2 - g_ref).max() <= GRAD_RTOL * np.abs(g_ref).max()
    qgd.clear_cache()


def test_rabi_swap_closed_form(qgd):
    """|Omega| = 1/2 for tf = pi is a SWAP (rabi_oscillator.jl:1-6): infidelity -> 0 at order 8."""
    prob = qgd.construct_rabi_prob(tf=np.pi, nsteps=50)
    ctrl = qgd.GRAPEControl(1, prob.tf)
    target = np.array([[0, 1], [1, 0]], dtype=complex) * (-1j)
    val = qgd.infidelity(prob, ctrl, np.array([0.5, 0.0]), target, order=8)
    assert abs(val) < 1e-12
    psi = qgd.eval_forward(prob, ctrl, np.array([0.5, 0.0]), order=8)
    assert np.abs(np.abs(psi[:, -1, :]) - np.array([[0, 1], [1, 0]])).max() < 1e-12
    qgd.clear_cache()


def test_full_size_properties(qgd):
    """cnot3 at BASELINE's full grid (N=64, 8 columns, order 8, tf=550): size-independent
    properties.  (1) Observed order of accuracy: the loss of orthonormality of the propagated
    basis (zero for the exact flow) at dt=0.5 vs dt=1 must show order 8 +- 1 (the reference's
    convergence criterion is nominal +- 0.5 in the asymptotic regime,
    test/ConvergenceTests/forward_convergence.jl:47-65).  (2) history_precomputed reuse returns
    the same scalars.  (3) The gradient matches a central-difference directional derivative of
    infidelity + guard penalty."""
    devs = {}
    for nsteps in (550, 1100):
        prob, target = qgd.cnot3_problem(nsteps=nsteps, tf=550.0)
        ctrl = cases.cnot3_controls(qgd, prob)
        npar = qgd.get_number_of_control_parameters(ctrl)
        pcof = (np.random.default_rng(0).random(npar) - 0.5) * 2 * np.pi * 0.005
        dp = qgd.device_problem(prob, 8)
        dp.set_controls(ctrl)
        dp.set_target(target)
        hist = np.zeros((128, 5, nsteps + 1, 8), order="F")
        out3 = dp.eval_forward(pcof, hist)
        psiN = hist[:64, 0, -1, :] + 1j * hist[64:, 0, -1, :]
        devs[nsteps] = np.abs(psiN.conj().T @ psiN - np.eye(8)).max()
        if nsteps == 550:
            grad, out3b = dp.discrete_adjoint(pcof, history_precomputed=True)
            assert np.allclose(out3, out3b, rtol=0, atol=1e-14)
            direction = np.random.default_rng(1).standard_normal(npar)
            direction /= np.linalg.norm(direction)
            eps = 1e-6

            def obj(p):
                a, b, g = dp.eval_forward(p)
                return 1 - (a * a + b * b) / prob.N_ess_levels ** 2 + g

            fd = (obj(pcof + eps * direction) - obj(pcof - eps * direction)) / (2 * eps)
            assert abs(fd - grad @ direction) <= 1e-6 * max(1.0, abs(fd))
    observed_order = np.log2(devs[550] / devs[1100])
    assert 7.0 <= observed_order <= 9.0, (devs, observed_order)
    qgd.clear_cache()


@pytest.mark.parametrize("which,nsteps,world", [("cnot2", 100, 2), ("cnot2", 100, 4), ("guarded", 90, 3), ("cnot3", 64, 2),
                                                ("cnot3", 96, 8), ("cnot3", 400, 2), ("cnot3", 550, 3), ("guarded", 420, 2)])
def test_time_partitioned_matches_single_gpu(qgd, which, nsteps, world):
    """The multi-GPU algorithm (time windows per rank, two all-gathers and one all-reduce per
    evaluation) with all ranks inside this process on the one GPU: gradient and scalars must equal
    the unpartitioned evaluation to rounding."""
    import torch
    prob, ctrl, pcof, target = getattr(cases, which + "_case")(qgd, nsteps=nsteps, tf=float(nsteps) / 2)
    order = 8 if which != "guarded" else 6
    dp = qgd.DeviceProblem(prob, order)
    dp.set_controls(ctrl)
    dp.set_target(target)
    g_ref, o_ref = dp.discrete_adjoint(pcof)
    dp.close()
    stream = torch.cuda.current_stream().cuda_stream
    backs = [qgd.DeviceBackend(prob, order, ctrl, target, r, world, device=0, stream=stream) for r in range(world)]
    cover = sorted((b.n_lo, b.n_hi) for b in backs)
    assert cover[0][0] == 0 and cover[-1][1] == nsteps
    assert all(cover[i][1] == cover[i + 1][0] for i in range(world - 1))      # windows share their end points
    results = qgd.LocalGroup(backs).discrete_adjoint(pcof)
    for g, o in results:
        assert np.abs(g - g_ref).max() <= 1e-12 * np.abs(g_ref).max()
        assert np.abs(o - o_ref).max() <= 1e-12
    again = qgd.LocalGroup(backs).discrete_adjoint(pcof)      # every rank adds its guard partials in a fixed order: the same bits again
    for (g, o), (g2, o2) in zip(results, again):
        assert np.array_equal(g, g2) and np.array_equal(np.asarray(o), np.asarray(o2))
    for b in backs:
        b.close()


@pytest.mark.parametrize("which", ["cnot2", "cnot3", "guarded"])
def test_graph_replay_matches_plain_launches(qgd, which, monkeypatch):
    """With QGD_GRAPH=1 a handle replays the captured launch sequence (hipGraph) from the third full evaluation
    on.  Replays with new pcof must give what a fresh handle (plain launches) gives, and setters must drop the graph."""
    monkeypatch.setenv("QGD_GRAPH", "1")
    prob, ctrl, pcof, target = getattr(cases, which + "_case")(qgd)
    order = 8 if which != "guarded" else 6
    rng = np.random.default_rng(3)
    pcs = [pcof * (1.0 + 0.3 * rng.standard_normal(len(pcof))) for _ in range(6)]
    dp = qgd.DeviceProblem(prob, order); dp.set_controls(ctrl); dp.set_target(target)
    got = [dp.discrete_adjoint(p) for p in pcs]
    monkeypatch.delenv("QGD_GRAPH")          # handles created from here on launch plainly
    for i in (0, 3, 5):
        fresh = qgd.DeviceProblem(prob, order); fresh.set_controls(ctrl); fresh.set_target(target)
        g, o = fresh.discrete_adjoint(pcs[i])
        fresh.close()
        assert np.abs(got[i][0] - g).max() <= 1e-13 * np.abs(g).max(), i
        assert np.abs(np.asarray(got[i][1]) - np.asarray(o)).max() <= 1e-13, i
    # a setter in between: new target -> the graph is rebuilt, results follow the new target
    target2 = np.roll(target, 1, axis=1)
    dp.set_target(target2)
    again = [dp.discrete_adjoint(p) for p in pcs[:4]]
    fresh = qgd.DeviceProblem(prob, order); fresh.set_controls(ctrl); fresh.set_target(target2)
    g, o = fresh.discrete_adjoint(pcs[3])
    fresh.close()
    assert np.abs(again[3][0] - g).max() <= 1e-13 * np.abs(g).max()
    # history_precomputed after a replayed evaluation reuses its forward history
    g2, _ = dp.discrete_adjoint(pcs[3], history_precomputed=True)
    assert np.abs(g2 - g).max() <= 1e-13 * np.abs(g).max()
    dp.close()
    qgd.clear_cache()


def test_row_packed_ell_slots(qgd, orc, monkeypatch):
    """Row-packed ELL slots -- what an operator pattern that is not banded gets -- instead of the diagonal-ordered ones of
    the banded a +- a^dagger patterns (QGD_PATHS=ell_row_packed forces them here): the same gradient (cnot3 order 8, guarded
    two-qutrit problem order 6) as the numpy statement of the algorithm."""
    monkeypatch.setenv("QGD_PATHS", "ell_row_packed")
    qgd.clear_cache()
    for which, order in (("cnot3", 8), ("guarded", 6)):
        prob, ctrl, pcof, target = getattr(cases, which + "_case")(qgd)
        Gp, Gq, off = qgd.control_basis(ctrl, prob.nsteps, prob.tf, order // 2)
        ref = pp.evaluate(prob, Gp, Gq, off, pcof, target, order)
        cases.oracle_pins(orc, prob, ctrl, pcof, target, order, ref)
        dp = qgd.DeviceProblem(prob, order); dp.set_controls(ctrl); dp.set_target(target)
        assert dp.operator_path()[0] == "sparse"
        grad, _ = dp.discrete_adjoint(pcof)
        dp.close()
        assert np.abs(grad - ref["grad"]).max() <= 1e-11 * np.abs(ref["grad"]).max(), which
    qgd.clear_cache()


@pytest.mark.parametrize("n_basis,force_copy", [(10, True), (300, False)])
def test_pcof_upload_paths(qgd, orc, n_basis, force_copy, monkeypatch):
    """pcof travels in the kernel arguments of k_tables when it fits (<= 448 coefficients) and through a device
    copy otherwise (here 2 x 600) or with QGD_PATHS=pcof_copy: both against the numpy statement of the algorithm."""
    if force_copy:
        monkeypatch.setenv("QGD_PATHS", "pcof_copy")
    prob, target = qgd.cnot2_problem(nsteps=24, tf=24.0)
    ctrl = [qgd.GeneralBSplineControl(2, n_basis, prob.tf) for _ in range(prob.N_operators)]
    npar = qgd.get_number_of_control_parameters(ctrl)
    assert (npar > 448) == (not force_copy)
    pcof = 0.05 * (0.5 - np.random.default_rng(4).random(npar))
    order = 6
    Gp, Gq, off = qgd.control_basis(ctrl, prob.nsteps, prob.tf, order // 2)
    ref = pp.evaluate(prob, Gp, Gq, off, pcof, target, order)
    cases.oracle_pins(orc, prob, ctrl, pcof, target, order, ref)
    dp = qgd.DeviceProblem(prob, order); dp.set_controls(ctrl); dp.set_target(target)
    for scale in (1.0, 0.5):            # two different vectors through the same handle
        ref = pp.evaluate(prob, Gp, Gq, off, scale * pcof, target, order)
        cases.oracle_pins(orc, prob, ctrl, scale * pcof, target, order, ref)
        grad, _ = dp.discrete_adjoint(scale * pcof)
        assert np.abs(grad - ref["grad"]).max() <= 1e-11 * np.abs(ref["grad"]).max()
    dp.close()
    qgd.clear_cache()


@pytest.mark.parametrize("world", [2, 3])
def test_time_partitioned_large_n(qgd, world):
    """Time windows with the large-N kernels (GEMM tiles, 32-column chain tiles, blocked inverse): N=100,
    32 columns, order 12, 60 steps split over 2 and 3 ranks on the one GPU."""
    import torch
    prob, ctrl, pcof, target = cases.synthetic_case(qgd, N=100, c=32, nsteps=60, tf=0.6)
    order = 12
    dp = qgd.DeviceProblem(prob, order)
    dp.set_controls(ctrl); dp.set_target(target)
    g_ref, o_ref = dp.discrete_adjoint(pcof)
    dp.close()
    stream = torch.cuda.current_stream().cuda_stream
    backs = [qgd.DeviceBackend(prob, order, ctrl, target, r, world, device=0, stream=stream) for r in range(world)]
    for g, o in qgd.LocalGroup(backs).discrete_adjoint(pcof):
        assert np.abs(g - g_ref).max() <= 1e-12 * np.abs(g_ref).max()
        assert np.abs(o - o_ref).max() <= 1e-12 * max(1.0, np.abs(o_ref).max())
    for b in backs:
        b.close()


@pytest.mark.parametrize("order", [2, 4, 6, 10, 12])
def test_cnot3_gradient_all_orders(qgd, orc, order):
    """Every instantiation of the N=64 fast-path kernels (fused L/R build and fused gradient kernel
    for orders 2..10, generic kernels at order 12) against the numpy statement of the algorithm,
    which tests/test_oracle.py ties to the reference-structured oracle."""
    prob, ctrl, pcof, target = cases.cnot3_case(qgd, nsteps=30, tf=15.0)
    Gp, Gq, off = qgd.control_basis(ctrl, prob.nsteps, prob.tf, order // 2)
    ref = pp.evaluate(prob, Gp, Gq, off, pcof, target, order)
    cases.oracle_pins(orc, prob, ctrl, pcof, target, order, ref)
    shape = (128, 1 + order // 2, 31, 8)
    hist = np.zeros(shape, order="F"); lam = np.zeros(shape, order="F"); forcing = np.zeros((128, 31, 8), order="F")
    grad = np.zeros(len(pcof))
    qgd.discrete_adjoint_(grad, hist, lam, forcing, prob, ctrl, pcof, target, order=order)
    assert close(hist, pp.history_real(ref["ws"]), 1e-12)
    assert np.abs(grad - ref["grad"]).max() <= 1e-11 * np.abs(ref["grad"]).max()
    grad2 = qgd.discrete_adjoint(prob, ctrl, pcof, target, order=order)       # without history copy-out
    assert np.abs(grad2 - ref["grad"]).max() <= 1e-11 * np.abs(ref["grad"]).max()
    qgd.clear_cache()


def test_optimize_gate_rabi_swap(qgd, tmp_path):
    """The reference's end-to-end test (test/OptimizationTests/optimization_rabi_osc_SWAP.jl:18-39):
    from 25 starts around the analytic optimum, optimize_gate(order=8, ridge 0) must return
    pcof ~ [0.5, 0] with rtol 5e-4 (Julia isapprox on the vector norm).  One run also writes the result file of
    update_jld2 (src/ipopt_optimal_control.jl:223-241) and reads it back."""
    prob = qgd.construct_rabi_prob(tf=np.pi, nsteps=20)
    control = qgd.GRAPEControl(1, prob.tf)
    target = np.array([[0.0, 1.0], [1.0, 0.0]])
    opt = np.array([0.5, 0.0])
    for p0 in np.linspace(0.4, 0.6, 5):
        for q0 in np.linspace(-0.1, 0.1, 5):
            hist = qgd.optimize_gate(prob, control, np.array([p0, q0]), target, order=8,
                                     ridge_penalty_strength=0, print_level=0)
            final = hist.pcof[-1]
            # |Omega| = 1/2 is a circle of optima; the reference's tolerance is on the distance to [0.5, 0]
            assert abs(np.hypot(*final) - 0.5) < 1e-4, (p0, q0, final)
            assert hist.analytic_obj_value[-1] < 1e-6
            if abs(q0) < 1e-12:
                assert np.linalg.norm(final - opt) <= 5e-4 * 0.5, (p0, q0, final)
    if qgd.jld2io.available():
        f = tmp_path / "rabi_swap.jld2"
        hist = qgd.optimize_gate(prob, control, np.array([0.45, 0.05]), target, order=8, ridge_penalty_strength=0,
                                 print_level=0, filename=f)
        back = qgd.read_optimization_history(f)
        assert len(back) == len(hist) and np.array_equal(back.pcof[-1], hist.pcof[-1]) and back.infidelity == hist.infidelity
        setup = qgd.jld2io.load(f)["Setup"]
        assert setup["order"] == 8 and np.array_equal(setup["target"], target) and setup["schrodinger_prob"]["nsteps"] == 20
    qgd.clear_cache()


@pytest.mark.parametrize("which,order", [("cnot2", 4), ("guarded", 8), ("cnot3", 8)])
def test_eval_adjoint_vs_oracle(qgd, orc, which, order):
    """eval_adjoint from a given terminal condition with forcing (forward_evolution.jl:352-483)."""
    prob, ctrl, pcof, target = getattr(cases, which + "_case")(qgd)
    rng = np.random.default_rng(8)
    n2, c, nt = prob.real_system_size, prob.N_initial_conditions, prob.nsteps + 1
    term = rng.standard_normal((n2, c))
    forcing = 0.1 * rng.standard_normal((n2, nt, c))
    ref = orc.eval_adjoint(prob, ctrl, pcof, term, order=order, forcing=forcing)
    got = qgd.eval_adjoint(prob, ctrl, pcof, term, order=order, forcing=forcing)
    assert close(got[:, 0], ref[:, 0], 1e-10)
    got0 = qgd.eval_adjoint(prob, ctrl, pcof, np.vstack([prob.u0, prob.v0]), order=order)      # as regression.jl:46-49
    ref0 = orc.eval_adjoint(prob, ctrl, pcof, np.vstack([prob.u0, prob.v0]), order=order)
    assert close(got0[:, 0], ref0[:, 0], 1e-10)
    qgd.clear_cache()


@pytest.mark.parametrize("order", [2, 4, 8, 10])
def test_lambda_history_derivative_columns(qgd, orc, order):
    """lambda_history[:, 1:, :, :] as the reference leaves it (forward_evolution.jl:427-433, :471-480: the adjoint
    derivatives of compute_adjoint_derivatives!, hermite.jl:284-305, with the controls at t_{n-1}, t_1 for n = 1, time
    index 0 never written) -- the opt-in output of qgd_set_lambda_derivatives against the oracle's tree recursion, through
    discrete_adjoint! and through eval_adjoint; and the default, which leaves those columns zero."""
    for name, prob, ctrl, pcof, target in cases.gradient_cases(qgd):
        g_ref, h_ref, lam_ref, f_ref, _ = orc.discrete_adjoint(prob, ctrl, pcof, target, order=order, return_all=True)
        lam = np.full(h_ref.shape, np.nan, order="F")
        grad = np.zeros_like(g_ref)
        qgd.discrete_adjoint_(grad, None, lam, None, prob, ctrl, pcof, target, order=order, lambda_derivatives=True)
        assert np.abs(lam_ref[:, 1:]).max() > 0 and not np.abs(lam_ref[:, :, 0]).any()
        for j in range(order // 2 + 1):
            assert np.abs(lam[:, j] - lam_ref[:, j]).max() <= 1e-10 * max(1.0, np.abs(lam_ref[:, j]).max()), (name, order, j)
        assert np.abs(grad - g_ref).max() <= GRAD_RTOL * np.abs(g_ref).max()
        lam0 = np.full(h_ref.shape, np.nan, order="F")
        qgd.discrete_adjoint_(grad, None, lam0, None, prob, ctrl, pcof, target, order=order)
        assert not np.abs(lam0[:, 1:]).any() and np.array_equal(lam0[:, 0], lam[:, 0])
    prob, ctrl, pcof, target = cases.guarded_case(qgd)
    rng = np.random.default_rng(order)
    n2, c, nt = prob.real_system_size, prob.N_initial_conditions, prob.nsteps + 1
    term = rng.standard_normal((n2, c))
    forcing = 0.1 * rng.standard_normal((n2, nt, c))
    ref = orc.eval_adjoint(prob, ctrl, pcof, term, order=order, forcing=forcing)
    got = qgd.eval_adjoint(prob, ctrl, pcof, term, order=order, forcing=forcing, lambda_derivatives=True)
    for j in range(order // 2 + 1):
        assert np.abs(got[:, j] - ref[:, j]).max() <= 1e-10 * max(1.0, np.abs(ref[:, j]).max()), j
    qgd.clear_cache()


def test_lambda_history_derivative_columns_cnot3(qgd, orc):
    """The same on the headline problem (sparse-operator path, N = 64, order 8), with a registered lambda_history."""
    prob, ctrl, pcof, target = cases.cnot3_case(qgd)
    orc.set_converged_terminal(True)
    try:
        g_ref, h_ref, lam_ref, f_ref, _ = orc.discrete_adjoint(prob, ctrl, pcof, target, order=8, return_all=True)
    finally:
        orc.set_converged_terminal(False)
    dp = qgd.DeviceProblem(prob, 8)
    dp.set_controls(ctrl); dp.set_target(target)
    lam = dp.pin(np.zeros(h_ref.shape, order="F"))
    g0, _ = dp.discrete_adjoint(pcof, False, None, lam)          # default first: zeros in the derivative columns
    assert not np.abs(lam[:, 1:]).any()
    dp.set_lambda_derivatives(True)
    g1, _ = dp.discrete_adjoint(pcof, False, None, lam)
    for j in range(5):
        assert np.abs(lam[:, j] - lam_ref[:, j]).max() <= 1e-10 * max(1.0, np.abs(lam_ref[:, j]).max()), j
    dp.set_lambda_derivatives(False)                              # and back: the registered array is cleared again
    g2, _ = dp.discrete_adjoint(pcof, False, None, lam)
    assert not np.abs(lam[:, 1:]).any() and close(lam[:, 0], lam_ref[:, 0], 1e-10)
    assert np.abs(g1 - g_ref).max() <= GRAD_RTOL * np.abs(g_ref).max() and np.abs(g1 - g0).max() <= 1e-13 * np.abs(g0).max()
    del lam
    dp.close()


@pytest.mark.parametrize("generic_chain", [False, True])
def test_large_n_fallback_paths(qgd, orc, generic_chain, monkeypatch):
    """N=100 (padded to 112), 20 columns (3 groups), 4 control operators, order 12 -- vs the numpy statement
    of the algorithm.  generic_chain: the chain kernel that sizes beyond Np = 640 take (QGD_PATHS=chain_generic) on this shape."""
    if generic_chain:
        monkeypatch.setenv("QGD_PATHS", "chain_generic")
    qgd.clear_cache()
    prob, ctrl, pcof, target = cases.synthetic_case(qgd)
    order = 12
    Gp, Gq, off = qgd.control_basis(ctrl, prob.nsteps, prob.tf, order // 2)
    ref = pp.evaluate(prob, Gp, Gq, off, pcof, target, order)
    cases.oracle_pins(orc, prob, ctrl, pcof, target, order, ref)
    shape = (200, 7, 25, 20)
    hist = np.zeros(shape, order="F"); lam = np.zeros(shape, order="F"); forcing = np.zeros((200, 25, 20), order="F")
    grad = np.zeros(len(pcof))
    qgd.discrete_adjoint_(grad, hist, lam, forcing, prob, ctrl, pcof, target, order=order)
    assert close(hist, pp.history_real(ref["ws"]), 1e-11)
    assert np.abs(grad - ref["grad"]).max() <= 1e-10 * np.abs(ref["grad"]).max()
    qgd.clear_cache()


@pytest.mark.parametrize("c", [8, 16, 32])
def test_large_n_gemm_kernels(qgd, orc, c):
    """N=100 (padded to 112: partial row tiles) with 8, 16 and 32 columns: the three tile shapes of the
    GEMM-style large-N kernels of qgd_k_dense.hip (<4,1>, <4,2>, <2,4>: fragment-ordered A_d / D_j,
    w_j = D_j w_0, level-by-level reverse sweep), order 12 and order 4, against the numpy statement."""
    for order, nsteps in ((12, 10), (4, 14)):
        prob, ctrl, pcof, target = cases.synthetic_case(qgd, N=100, c=c, nsteps=nsteps, tf=0.3)
        Gp, Gq, off = qgd.control_basis(ctrl, prob.nsteps, prob.tf, order // 2)
        ref = pp.evaluate(prob, Gp, Gq, off, pcof, target, order)
        cases.oracle_pins(orc, prob, ctrl, pcof, target, order, ref)
        dp = qgd.DeviceProblem(prob, order)
        dp.set_controls(ctrl); dp.set_target(target)
        grad, _ = dp.discrete_adjoint(pcof)
        assert np.abs(grad - ref["grad"]).max() <= 1e-11 * np.abs(ref["grad"]).max(), order
        hist = np.zeros((200, order // 2 + 1, nsteps + 1, c), order="F")
        qgd.eval_forward_(hist, prob, ctrl, pcof, order=order)
        assert close(hist, pp.history_real(ref["ws"]), 1e-12), order
    qgd.clear_cache()


@pytest.mark.parametrize("N,c,order,n_ops,nsteps", [(65, 1, 2, 2, 9), (72, 9, 4, 1, 13), (96, 40, 6, 3, 11), (130, 17, 16, 2, 5),
                                                  (200, 3, 2, 5, 7), (113, 24, 8, 3, 30), (255, 5, 4, 3, 6), (290, 12, 6, 2, 5)])
def test_large_n_shape_sweep(qgd, orc, N, c, order, n_ops, nsteps):
    """Odd sizes for the N > 64 kernels: partial row tiles and column groups, 1..40 columns, orders 2..16, 1..5
    control operators, N above the 288 limit of the blocked inverse / LDS chain -- history and gradient against
    the numpy statement of the algorithm (scripts/shape_sweep.py prints the errors: 1e-15 / 1e-13)."""
    prob, ctrl, pcof, target = cases.synthetic_case(qgd, N=N, c=c, n_ops=n_ops, nsteps=nsteps, tf=0.02 * nsteps, seed=N + c)
    Gp, Gq, off = qgd.control_basis(ctrl, prob.nsteps, prob.tf, order // 2)
    ref = pp.evaluate(prob, Gp, Gq, off, pcof, target, order)
    cases.oracle_pins(orc, prob, ctrl, pcof, target, order, ref)
    dp = qgd.DeviceProblem(prob, order); dp.set_controls(ctrl); dp.set_target(target)
    grad, _ = dp.discrete_adjoint(pcof)
    hist = np.zeros((2 * N, order // 2 + 1, nsteps + 1, c), order="F")
    qgd.eval_forward_(hist, prob, ctrl, pcof, order=order)
    dp.close()
    assert close(hist, pp.history_real(ref["ws"]), 1e-12)
    assert np.abs(grad - ref["grad"]).max() <= 1e-11 * np.abs(ref["grad"]).max()
    qgd.clear_cache()


def test_edge_no_controls_is_pade(qgd):
    """Empty control set (N_operators = 0): the sweep is the diagonal Pade approximant of exp(A dt)
    applied nsteps times -- closed form, no oracle needed.  Also exercises N=5 (padding), 3 columns."""
    import math
    rng = np.random.default_rng(21)
    N, c, nsteps, tf = 5, 3, 7, 0.9
    S = rng.standard_normal((N, N)); S = S + S.T
    K = rng.standard_normal((N, N)); K = K - K.T
    U0 = rng.standard_normal((N, c)) + 1j * rng.standard_normal((N, c))
    prob = qgd.SchrodingerProb(S, K, [], [], U0.real, U0.imag, None, tf, nsteps, N)
    Ac = K - 1j * S
    dt = tf / nsteps
    for order in (2, 6, 12, 16):
        m = order // 2
        cj = [math.factorial(m) * math.factorial(2 * m - j) / (math.factorial(2 * m) * math.factorial(m - j)) for j in range(m + 1)]
        num = sum(cj[j] * np.linalg.matrix_power(Ac * dt, j) / math.factorial(j) for j in range(m + 1))
        den = sum(cj[j] * np.linalg.matrix_power(-Ac * dt, j) / math.factorial(j) for j in range(m + 1))
        step = np.linalg.solve(den, num)
        psi = qgd.eval_forward(prob, [], np.zeros(0), order=order)
        ref = U0.copy()
        for n in range(nsteps):
            ref = step @ ref
            assert np.abs(psi[:, n + 1, :] - ref).max() < 1e-12 * max(1.0, np.abs(ref).max()), (order, n)
    qgd.clear_cache()


@pytest.mark.parametrize("nsteps", [1, 2, 3])
def test_edge_tiny_grids(qgd, orc, nsteps):
    """One, two and three timesteps (no scan blocks, adjoint loop of length 0 and 1), order 2 and 16."""
    prob = qgd.construct_rabi_prob(tf=0.3, nsteps=nsteps, gmres_abstol=1e-15, gmres_reltol=1e-15)
    ctrl = qgd.FortranBSplineControl(3, 6, prob.tf)
    pcof = np.random.default_rng(3).random(ctrl.N_coeff)
    target = cases.rand_target(prob)
    for order in (2, 16):
        g_ref, h_ref, lam_ref, f_ref, _ = orc.discrete_adjoint(prob, ctrl, pcof, target, order=order, return_all=True)
        hist = np.zeros(h_ref.shape, order="F")
        grad = np.zeros_like(g_ref)
        qgd.discrete_adjoint_(grad, hist, None, None, prob, ctrl, pcof, target, order=order)
        assert close(hist, h_ref), (nsteps, order)
        assert np.abs(grad - g_ref).max() <= GRAD_RTOL * np.abs(g_ref).max(), (nsteps, order)
    qgd.clear_cache()


def test_error_behaviour_on_device(qgd):
    """Call-order and argument errors surface as error codes with messages, never as silent fallbacks."""
    prob, ctrl, pcof, target = cases.cnot2_case(qgd, nsteps=8, tf=8.0)
    dp = qgd.DeviceProblem(prob, 4)
    with pytest.raises(qgd._lib.QGDError) as e:
        dp.eval_forward(pcof)                           # pcof without a control basis
    assert e.value.code == qgd._lib.QGD_ERR_STATE
    dp.set_controls(ctrl)
    with pytest.raises(qgd._lib.QGDError) as e:
        dp.discrete_adjoint(pcof)                       # gradient without a target
    assert e.value.code == qgd._lib.QGD_ERR_STATE
    with pytest.raises(qgd._lib.QGDError) as e:
        dp.eval_forward(pcof[:-1])                      # wrong pcof length
    assert e.value.code == qgd._lib.QGD_ERR_ARGUMENT
    dp.set_target(target)
    with pytest.raises(qgd._lib.QGDError) as e:
        dp.discrete_adjoint(pcof, history_precomputed=True)   # nothing to reuse yet
    assert e.value.code == qgd._lib.QGD_ERR_STATE
    dp.close()
    bad = prob.copy(); bad.sym_operators = [o + np.triu(np.ones_like(o), 1) for o in bad.sym_operators]
    with pytest.raises(qgd._lib.QGDError) as e:
        qgd.DeviceProblem(bad, 4)                       # asymmetric "symmetric" operator: ArgumentError
    assert e.value.code == qgd._lib.QGD_ERR_ARGUMENT and "not symmetric" in str(e.value)


@pytest.mark.parametrize("which,order", [("cnot3", 2), ("cnot3", 4), ("cnot3", 8), ("cnot3", 10), ("cnot3", 12),
                                         ("cnot3", 14), ("cnot3", 16), ("cnot2", 6), ("guarded", 8)])
def test_sparse_and_dense_operator_paths_agree(qgd, orc, which, order):
    """The ELL (sparse-operator) kernels and the dense fp64 MFMA kernels build the same L_n, R_n and
    the same gradient, and both match the numpy statement of the algorithm."""
    kw = dict(nsteps=24, tf=12.0) if which == "cnot3" else {}
    prob, ctrl, pcof, target = getattr(cases, which + "_case")(qgd, **kw)
    Gp, Gq, off = qgd.control_basis(ctrl, prob.nsteps, prob.tf, order // 2)
    ref = pp.evaluate(prob, Gp, Gq, off, pcof, target, order)
    cases.oracle_pins(orc, prob, ctrl, pcof, target, order, ref)
    out = {}
    for path in ("sparse", "dense"):
        dp = qgd.DeviceProblem(prob, order)
        dp.set_operator_path(path)
        assert dp.operator_path()[0] == path
        dp.set_controls(ctrl); dp.set_target(target)
        grad, out3 = dp.discrete_adjoint(pcof)
        L, R = dp.intermediate("L"), dp.intermediate("R")
        assert np.abs(L - ref["L"]).max() < 1e-13 and np.abs(R - ref["R"]).max() < 1e-13, path
        assert np.abs(grad - ref["grad"]).max() <= 1e-11 * np.abs(ref["grad"]).max(), path
        out[path] = (grad, out3, L, R)
        dp.close()
    gs, gd = out["sparse"][0], out["dense"][0]
    assert np.abs(gs - gd).max() <= 1e-12 * np.abs(gd).max()
    assert np.abs(out["sparse"][2] - out["dense"][2]).max() < 1e-14


def test_operator_path_selection(qgd):
    """cnot3 (drift diagonal + a_k +/- a_k^dagger on three subsystems) has 7 entries per row and 2 per
    control operator -> sparse; a random dense problem stays on the MFMA kernels and refuses 'sparse'."""
    import os
    if "dense_ops" in os.environ.get("QGD_PATHS", ""):
        pytest.skip("QGD_PATHS=dense_ops overrides the automatic choice")
    prob, ctrl, pcof, target = cases.cnot3_case(qgd, nsteps=4, tf=4.0)
    dp = qgd.DeviceProblem(prob, 8)
    assert dp.operator_path() == ("sparse", 7, 2)
    dp.set_operator_path("dense"); assert dp.operator_path()[0] == "dense"
    dp.set_operator_path("auto"); assert dp.operator_path()[0] == "sparse"
    dp.close()
    prob, ctrl, pcof, target = cases.synthetic_case(qgd, N=40, c=4, n_ops=2, nsteps=4)
    dp = qgd.DeviceProblem(prob, 4)
    assert dp.operator_path()[0] == "dense"
    with pytest.raises(qgd._lib.QGDError) as e:
        dp.set_operator_path("sparse")
    assert e.value.code == qgd._lib.QGD_ERR_UNSUPPORTED
    dp.close()


@pytest.mark.parametrize("N,c,order", [(20, 4, 6), (40, 12, 4), (64, 8, 8)])
def test_mid_size_dense_problems(qgd, orc, N, c, order):
    """Random dense problems padded to 32, 48 and 64 rows: the register-blocked inverses, the
    team-pipelined sweeps at every compiled size and the dense MFMA operator kernels, against the
    numpy statement of the algorithm."""
    prob, ctrl, pcof, target = cases.synthetic_case(qgd, N=N, c=c, n_ops=2, nsteps=70, tf=0.7)
    Gp, Gq, off = qgd.control_basis(ctrl, prob.nsteps, prob.tf, order // 2)
    ref = pp.evaluate(prob, Gp, Gq, off, pcof, target, order)
    cases.oracle_pins(orc, prob, ctrl, pcof, target, order, ref)
    shape = (2 * N, 1 + order // 2, prob.nsteps + 1, c)
    hist = np.zeros(shape, order="F"); lam = np.zeros(shape, order="F"); forcing = np.zeros((2 * N, prob.nsteps + 1, c), order="F")
    grad = np.zeros(len(pcof))
    qgd.discrete_adjoint_(grad, hist, lam, forcing, prob, ctrl, pcof, target, order=order)
    assert close(hist, pp.history_real(ref["ws"]), 1e-11)
    assert np.abs(grad - ref["grad"]).max() <= 1e-10 * np.abs(ref["grad"]).max()
    qgd.clear_cache()


def test_save_every_nsteps_and_convergence_report(qgd):
    """saveEveryNsteps (forward_evolution.jl:104,239-241) stores every k-th point of the same sweep, and
    get_histories (src/Tests/test_convergence.jl:20-146) reproduces the reference's convergence test
    (forward_convergence.jl:47-65): observed order = nominal +/- 0.5 from Richardson estimates."""
    prob = qgd.construct_rabi_prob(tf=1.0, nsteps=8, gmres_abstol=1e-15, gmres_reltol=1e-15)
    ctrl = qgd.FortranBSplineControl(16, 17, prob.tf)          # one polynomial piece: smooth to every order
    pcof = np.random.default_rng(5).random(ctrl.N_coeff)
    full = qgd.eval_forward(prob, ctrl, pcof, order=4)
    sub = qgd.eval_forward(prob, ctrl, pcof, order=4, saveEveryNsteps=4)
    assert sub.shape == (2, 3, prob.N_initial_conditions) and np.array_equal(sub, full[:, ::4, :])
    ret = qgd.get_histories(prob, ctrl, pcof, 5, orders=(2, 4, 6), base_nsteps=8, quiet=True)
    for order in (2, 4, 6):
        summary = ret[f"Order {order} (QGD)"]
        assert summary["nsteps"] == [8, 16, 32, 64, 128]
        assert all(h.shape == summary["histories"][0].shape for h in summary["histories"])
        obs = qgd.observed_orders(summary)
        good = [o for o, e in zip(obs, summary["richardson_errors"][2:]) if e > 1e-12]   # above rounding
        assert good and all(abs(o - order) < 0.6 for o in good), (order, obs, summary["richardson_errors"])
    qgd.clear_cache()


@pytest.mark.parametrize("order", [2, 4, 6, 8, 10])
def test_forced_gradient_reference_contract(qgd, orc, order):
    """The reference's own contract (test/GradientTests/compare_gradients.jl:47-65): the discrete adjoint
    and the forced (forward-sensitivity) gradient agree to rounding -- both on the device here -- and
    the device forced gradient matches the oracle's eval_grad_forced."""
    for name, prob, ctrl, pcof, target in cases.gradient_cases(qgd):
        g_adj = qgd.discrete_adjoint(prob, ctrl, pcof, target, order=order)
        g_for = qgd.eval_grad_forced(prob, ctrl, pcof, target, order=order)
        g_orc = orc.eval_grad_forced(prob, ctrl, pcof, target, order=order)
        scale = np.abs(g_adj).max()
        assert np.abs(g_for - g_adj).max() <= 1e-12 * scale, (name, order)
        assert np.abs(g_for - g_orc).max() <= 1e-10 * scale, (name, order)
        if order == 4:     # the third leg: centred differences (<= 1e-9 in the reference's test)
            g_fd = qgd.eval_grad_finite_difference(prob, ctrl, pcof, target, order=order)
            assert np.abs(g_fd - g_adj).max() <= 3e-9 * max(1.0, scale), (name, order)
    qgd.clear_cache()


@pytest.mark.parametrize("which,order,kw", [("cnot2", 4, {}), ("guarded", 8, {}), ("dense_guard", 6, {}),
                                            ("cnot3", 8, dict(nsteps=40, tf=40.0)), ("cnot3", 4, dict(nsteps=130, tf=65.0))])
def test_forced_gradient_configs(qgd, which, order, kw):
    """Forced gradient vs discrete adjoint on the benchmark-shaped problems (guard penalties, carrier
    controls, several scan blocks): agreement to 1e-11 relative."""
    prob, ctrl, pcof, target = getattr(cases, which + "_case")(qgd, **kw)
    g_adj = qgd.discrete_adjoint(prob, ctrl, pcof, target, order=order)
    g_for = qgd.eval_grad_forced(prob, ctrl, pcof, target, order=order)
    assert np.abs(g_for - g_adj).max() <= 1e-11 * np.abs(g_adj).max()
    qgd.clear_cache()


def test_legacy_controls_on_device(qgd):
    """BSplineControl (bcarrier2 layout), GeneralGRAPEControl and SinCosControl drive the device stepper:
    adjoint == forced to rounding and == central differences of the device objective."""
    prob = qgd.construct_rand_prob(6, 3, tf=1.2, nsteps=24, scale=0.4)
    ctrl = [qgd.BSplineControl(prob.tf, 5, [0.0, 1.9]), qgd.GeneralGRAPEControl(3, prob.tf, 2), qgd.SinCosControl(prob.tf, frequency=2.5)]
    rng = np.random.default_rng(12)
    pcof = 0.5 * rng.standard_normal(qgd.get_number_of_control_parameters(ctrl))
    target = cases.rand_target(prob)
    order = 6
    g_adj = qgd.discrete_adjoint(prob, ctrl, pcof, target, order=order)
    g_for = qgd.eval_grad_forced(prob, ctrl, pcof, target, order=order)
    assert np.abs(g_adj - g_for).max() <= 1e-12 * np.abs(g_adj).max()
    dp = qgd.device_problem(prob, order); dp.set_controls(ctrl); dp.set_target(target)

    def obj(p):
        a, b, g = dp.eval_forward(p)
        return 1 - (a * a + b * b) / prob.N_ess_levels ** 2 + g

    for l in rng.choice(len(pcof), 6, replace=False):
        e = np.zeros(len(pcof)); e[l] = 1e-6
        fd = (obj(pcof + e) - obj(pcof - e)) / 2e-6
        assert abs(fd - g_adj[l]) <= 2e-8 * max(1.0, np.abs(g_adj).max()), (l, fd, g_adj[l])
    qgd.clear_cache()


def test_legacy_controls_on_device_vs_oracle(qgd, orc):
    """SURVEY 8 row f3 on the DEVICE against the oracle (not device against device): the hard-coded quadratic B-spline
    under carrier waves (BSpline2Control, my_bspline_controls: bspline_control.jl:21-270, :52-64) at orders 2-6, and
    the reference's BSplineControl -- the product builds it as CarrierControl(BSpline2Control), the oracle evaluates it
    through its own restatement of bcarrier2 / gradbcarrier2 (bspline_backend.jl:381-955; derivative orders 0 and 1,
    i.e. Hermite order 2, all the reference implements).  Gradient against the oracle's discrete_adjoint AND
    eval_grad_forced (the reference's parity contract, test/GradientTests/compare_gradients.jl:47-65), state history with
    every stage derivative, lambda and the guard forcing: 1e-10 / 1e-11."""
    from oracle.oracle import BCarrier2Control
    prob = qgd.construct_rand_prob(5, 2, tf=1.5, nsteps=30, scale=0.5, gmres_abstol=1e-15, gmres_reltol=1e-15)
    target = cases.rand_target(prob)
    rng = np.random.default_rng(44)
    omega = [-2.5, 0.0, 1.7]
    cases_ = [([qgd.CarrierControl(qgd.BSpline2Control(6, prob.tf), omega), qgd.CarrierControl(qgd.BSpline2Control(4, prob.tf), [0.0, 3.1])],
               None, (2, 4, 6)),
              ([qgd.BSplineControl(prob.tf, 6, omega), qgd.BSplineControl(prob.tf, 5, [1.1])],
               [BCarrier2Control(prob.tf, 6, omega), BCarrier2Control(prob.tf, 5, [1.1])], (2,))]
    orc.set_converged_terminal(True)
    try:
        for dev_ctrl, orc_ctrl, orders in cases_:
            orc_ctrl = dev_ctrl if orc_ctrl is None else orc_ctrl
            pcof = 0.4 * rng.standard_normal(qgd.get_number_of_control_parameters(dev_ctrl))
            for order in orders:
                g_ref, h_ref, lam_ref, f_ref, _ = orc.discrete_adjoint(prob, orc_ctrl, pcof, target, order=order, return_all=True)
                gf_ref = orc.eval_grad_forced(prob, orc_ctrl, pcof, target, order=order)
                dp = qgd.DeviceProblem(prob, order); dp.set_controls(dev_ctrl); dp.set_target(target)
                hist, lam = np.zeros(h_ref.shape, order="F"), np.zeros(h_ref.shape, order="F")
                forcing = np.zeros(f_ref.shape, order="F")
                g, _ = dp.discrete_adjoint(pcof, False, hist, lam, forcing)
                gf = dp.eval_grad_forced(pcof)
                dp.close()
                scale = np.abs(g_ref).max()
                assert np.abs(g - g_ref).max() <= 1e-10 * scale, (type(orc_ctrl[0]).__name__, order)
                assert np.abs(g - gf_ref).max() <= 1e-10 * scale, (type(orc_ctrl[0]).__name__, order)
                assert np.abs(gf - gf_ref).max() <= 1e-10 * scale, (type(orc_ctrl[0]).__name__, order)
                assert close(hist, h_ref) and close(lam[:, 0], lam_ref[:, 0]) and close(forcing, f_ref), (type(orc_ctrl[0]).__name__, order)
    finally:
        orc.set_converged_terminal(False)


@pytest.mark.parametrize("which,order", [("cnot2", 4), ("guarded", 6), ("cnot3", 8)])
def test_eval_forward_with_forcing(qgd, orc, which, order):
    """eval_forward(...; forcing) (forward_evolution.jl:118-129,167-206): the whole derivative history and the
    objective scalars with a random forcing, against the oracle's forced forward sweep."""
    kw = dict(nsteps=40, tf=20.0) if which == "cnot3" else {}
    prob, ctrl, pcof, target = getattr(cases, which + "_case")(qgd, **kw)
    m = order // 2
    rng = np.random.default_rng(31)
    forcing = np.asfortranarray(0.3 * rng.standard_normal((prob.real_system_size, m, prob.nsteps + 1, prob.N_initial_conditions)))
    h_ref = orc.eval_forward(prob, ctrl, pcof, order=order, forcing=forcing)
    hist = np.zeros(h_ref.shape, order="F")
    qgd.eval_forward_(hist, prob, ctrl, pcof, order=order, forcing=forcing)
    assert close(hist, h_ref)
    psi = qgd.eval_forward(prob, ctrl, pcof, order=order, forcing=forcing)
    assert close(psi, h_ref[:prob.N_tot_levels, 0] + 1j * h_ref[prob.N_tot_levels:, 0])
    # and without forcing the same entry point reproduces the ordinary sweep
    plain = qgd.eval_forward(prob, ctrl, pcof, order=order, forcing=np.zeros_like(forcing))
    assert close(plain, qgd.eval_forward(prob, ctrl, pcof, order=order), 1e-13)
    # the three scalars too (overlaps and GUARD PENALTY: round 3's fixed-order sum dropped most of it on this entry point)
    dp = qgd.DeviceProblem(prob, order); dp.set_controls(ctrl); dp.set_target(target)
    s_plain = np.asarray(dp.eval_forward(pcof))
    s_zero = np.asarray(dp.eval_forward_forced(pcof, np.zeros_like(forcing)))
    assert np.abs(s_zero - s_plain).max() <= 1e-12 * max(1.0, np.abs(s_plain).max()), (s_zero, s_plain)
    s_forced = np.asarray(dp.eval_forward_forced(pcof, forcing))
    assert abs(s_forced[2] - orc.guard_penalty_real(prob, h_ref)) <= 1e-11 * max(1.0, abs(s_forced[2]))
    dp.close()
    qgd.clear_cache()


def test_dahlquist_and_rotating_frame_on_device(qgd):
    """The scalar test equation y' = lambda y (dahlquist_problem.jl) integrates to exp(lambda t) (N = 1: the
    smallest padded problem), and a rotating-frame qubit's adjoint gradient matches its forced gradient."""
    lam = 2.0j
    prob = qgd.dahlquist_problem(lam, initial_condition=0.6 - 0.8j)
    psi = qgd.eval_forward(prob, [], np.zeros(0), order=12)
    t = np.linspace(0, prob.tf, prob.nsteps + 1)
    assert np.abs(psi[0, :, 0] - (0.6 - 0.8j) * np.exp(lam * t)).max() < 1e-13
    q = qgd.rotating_frame_qubit(2, 2, tf=2.0, nsteps=40, detuning_frequency=0.3, self_kerr_coefficient=0.2)
    ctrl = qgd.BSplineControl(q.tf, 6, [0.0, 0.7])
    pcof = 0.2 * np.random.default_rng(2).standard_normal(ctrl.N_coeff)
    target = cases.rand_target(q)
    g_adj = qgd.discrete_adjoint(q, ctrl, pcof, target, order=6)
    g_for = qgd.eval_grad_forced(q, ctrl, pcof, target, order=6)
    assert np.abs(g_adj - g_for).max() <= 1e-12 * np.abs(g_adj).max()
    qgd.clear_cache()


def test_column_block_inverse_and_its_fallbacks(qgd, orc, monkeypatch):
    """N = 64: the column-block elimination of [L | R] (qgd_inverse_cb.h) and the two stages behind it.  (1) cnot3: every matrix
    is done by the first attempt (pivots on the diagonal) and gradient, L^-1 and P equal those of the fully pivoted 4-pivot-panel
    kernel (QGD_PATHS=inv_panels) to rounding.  (2) A drift that pairs the levels i <-> i^1 with dt*h at the zero of Re q_3(iy):
    the diagonal of L vanishes beside off-diagonal entries of modulus ~1 INSIDE the 16 x 16 diagonal tiles -- the diagonal
    attempt is given up, partial pivoting inside the tiles does every matrix.  (3) The same drift pairing i <-> i^32: the big
    entries lie outside the diagonal tiles, both column-block attempts are given up and the fully pivoted elimination does
    every matrix.  (2) and (3) against the numpy statement, itself tied to the oracle: the stages are exercised, not just
    present.  intermediate("repivoted") = matrices past the first attempt + 65536 * matrices past the second."""
    # the kernel has two instantiations (qgd_inverse_cb.h, ONE): 256 < matrices <= 768, where every workgroup of the launch is
    # resident at once, and the rest -- 40 and 800 steps run the second, 300 the first
    for nsteps in (40, 300, 800):
        prob, ctrl, pcof, target = cases.cnot3_case(qgd, nsteps=nsteps, tf=float(nsteps))
        res = {}
        for tag, env in (("blocks", {}), ("panels", {"QGD_PATHS": "inv_panels"})):
            for k_, v in env.items():
                monkeypatch.setenv(k_, v)
            dp = qgd.DeviceProblem(prob, 8)
            dp.set_controls(ctrl); dp.set_target(target)
            res[tag] = dp.discrete_adjoint(pcof) + (dp.intermediate("repivoted"), dp.intermediate("Linv"), dp.intermediate("P"))
            dp.close()
            for k_ in env:
                monkeypatch.delenv(k_)
        assert res["blocks"][2] == 0, nsteps
        assert np.abs(res["panels"][0] - res["blocks"][0]).max() <= 1e-12 * np.abs(res["blocks"][0]).max(), nsteps
        assert np.abs(res["panels"][3] - res["blocks"][3]).max() <= 1e-12 * np.abs(res["blocks"][3]).max(), nsteps
        assert np.abs(res["panels"][4] - res["blocks"][4]).max() <= 1e-12, nsteps
    order, nsteps, dt = 6, 12, 0.8 / 12
    for partner, stage in ((1, 1), (32, 2)):
        prob, ctrl, pcof, target = cases.synthetic_case(qgd, N=64, c=8, n_ops=2, nsteps=nsteps, tf=0.8, seed=9)
        H = np.zeros((64, 64))
        for i in range(64):
            H[i, i ^ partner] = np.sqrt(10.0) / dt
        prob.system_sym = np.asfortranarray(H + 0.1 * prob.system_sym)
        Gp, Gq, off = qgd.control_basis(ctrl, prob.nsteps, prob.tf, order // 2)
        ref = pp.evaluate(prob, Gp, Gq, off, 0.1 * pcof, target, order)
        if partner == 1:
            cases.oracle_pins(orc, prob, ctrl, 0.1 * pcof, target, order, ref)
        dp = qgd.DeviceProblem(prob, order)
        dp.set_controls(ctrl); dp.set_target(target)
        grad, _ = dp.discrete_adjoint(0.1 * pcof)
        count = int(dp.intermediate("repivoted"))
        P = dp.intermediate("P")
        # the handle remembers that the diagonal attempt failed for (more than a quarter of) its matrices: the next evaluation
        # starts with pivoting -- same counts, same result
        grad2, _ = dp.discrete_adjoint(0.1 * pcof)
        assert int(dp.intermediate("repivoted")) == count and np.abs(grad2 - grad).max() <= 1e-13 * np.abs(grad).max()
        dp.close()
        assert (count & 0xFFFF) == nsteps, (partner, count)
        assert (count >> 16) == (nsteps if stage == 2 else 0), (partner, count)
        assert np.abs(grad - ref["grad"]).max() <= 1e-10 * np.abs(ref["grad"]).max(), partner
        assert np.abs(P[:nsteps] - ref["P"]).max() <= 1e-11, partner


def test_non_finite_coefficients_end_and_leave_the_handle_usable(qgd):
    """A coefficient vector with a NaN / an infinity / 1e200 in it (a line search that overshoots): the evaluation ENDS, its
    results are non-finite, as the reference's would be, and the next evaluation of the same handle is bit for bit what it was
    before.  (Non-finite step matrices go through the diagonal attempt of the N = 64 inverse: `repivoted` stays 0.)"""
    nsteps = 40
    prob, ctrl, pcof, target = cases.cnot3_case(qgd, nsteps=nsteps, tf=float(nsteps))
    dp = qgd.DeviceProblem(prob, 8)
    dp.set_controls(ctrl); dp.set_target(target)
    g0, o0 = dp.discrete_adjoint(pcof)
    assert int(dp.intermediate("repivoted")) == 0
    for bad in (np.nan, np.inf, 1e200):
        p2 = pcof.copy(); p2[3] = bad
        g, o = dp.discrete_adjoint(p2)
        assert not np.isfinite(g).all() and not np.isfinite(o[0])
        g1, o1 = dp.discrete_adjoint(pcof)
        assert np.array_equal(g1, g0) and np.array_equal(np.asarray(o1), np.asarray(o0)), bad
        assert int(dp.intermediate("repivoted")) == 0
    dp.close()


def test_inverse_returns_to_the_diagonal_attempt(qgd):
    """One evaluation whose step matrices are finite but far from diagonally dominant (coefficients a thousand times too large: the
    diagonal attempt of the N = 64 inverse is given up for more than a quarter of them) makes the next 32 evaluations of the handle
    START with pivoting (qgd_k_build.hip: inverse_memory; `repivoted` counts every matrix while they do); then the diagonal attempt
    is tried again and, the matrices being what they were, stays.  The results agree to 1e-12 throughout (in-tile pivoting rounds
    differently from diagonal pivots: include/qgd.h documents the history dependence; within one regime the bits repeat)."""
    nsteps = 40
    prob, ctrl, pcof, target = cases.cnot3_case(qgd, nsteps=nsteps, tf=float(nsteps))
    dp = qgd.DeviceProblem(prob, 8)
    dp.set_controls(ctrl); dp.set_target(target)
    g0, _ = dp.discrete_adjoint(pcof)
    scale = None
    for s_ in (30.0, 100.0, 300.0, 1000.0):
        dp.discrete_adjoint(s_ * pcof)
        if 4 * (int(dp.intermediate("repivoted")) & 0xFFFF) > nsteps + 1:
            scale = s_
            break
    assert scale is not None
    counts = []
    for _ in range(36):
        g1, _ = dp.discrete_adjoint(pcof)
        counts.append(int(dp.intermediate("repivoted")) & 0xFFFF)
        assert np.abs(g1 - g0).max() <= 1e-12 * np.abs(g0).max()
    nmat = nsteps + 1 if dp.front_path_taken() else nsteps      # (the fused front inverts L at every time point, the general path at all but the first)
    assert counts[:32] == [nmat] * 32 and counts[32:] == [0] * 4, (scale, counts)
    dp.close()


@pytest.mark.parametrize("c", [20, 64])
def test_cnot3_many_columns(qgd, orc, c):
    """The sparse N=64 path with more initial conditions than one MFMA tile (the whole 64-dimensional basis, and a
    ragged 20): several column groups per time point in the sweeps, the gradient scalars summed over groups."""
    prob, ctrl, pcof, _ = cases.cnot3_case(qgd, nsteps=30, tf=30.0)
    U0 = np.eye(64, c)
    prob.u0 = np.asfortranarray(U0); prob.v0 = np.asfortranarray(np.zeros((64, c)))
    prob.N_initial_conditions = c
    target = cases.rand_target(prob, seed=c)
    order = 8
    Gp, Gq, off = qgd.control_basis(ctrl, prob.nsteps, prob.tf, order // 2)
    ref = pp.evaluate(prob, Gp, Gq, off, pcof, target, order)
    cases.oracle_pins(orc, prob, ctrl, pcof, target, order, ref)
    dp = qgd.DeviceProblem(prob, order)
    dp.set_controls(ctrl); dp.set_target(target)
    assert dp.operator_path()[0] == "sparse"
    hist = np.zeros((128, order // 2 + 1, prob.nsteps + 1, c), order="F")
    grad, out3 = dp.discrete_adjoint(pcof, False, hist)
    dp.close()
    assert close(hist, pp.history_real(ref["ws"]), 1e-11)
    assert np.abs(grad - ref["grad"]).max() <= 1e-10 * np.abs(ref["grad"]).max()
    a, b = ref["overlap"]          # the statement's overlap is sum conj(R) psi = <w,R> - i <w,T>  (T = [R_im; -R_re], infidelity.jl:13-17)
    assert abs(out3[0] - a) < 1e-11 and abs(out3[1] + b) < 1e-11 and abs(out3[2] - ref["guard"]) < 1e-11


@pytest.mark.parametrize("cost_type", ["Tracking", "Norm"])
def test_cost_types_vs_oracle(qgd, orc, cost_type):
    """cost_type = :Tracking / :Norm (eval_grad_discrete_adjoint.jl:26-35, eval_grad_forced.jl:155-165,
    eval_grad_finite_difference.jl:48-59): the terminal condition, lambda and the gradient against the oracle, the
    reference's three-way contract (adjoint = forced = centred differences) on the device, the cost scalar, the stored
    forward sweep, the fused and the stand-alone terminal kernel, the N > 64 path, and the error for anything else."""
    orc.set_converged_terminal(True)
    orc.set_cost_type(cost_type)
    try:
        todo = [(n, p, c, x, t, o) for (n, p, c, x, t) in cases.gradient_cases(qgd)[:4] for o in (2, 6)]
        todo += [("guarded",) + cases.guarded_case(qgd, nsteps=20, tf=10.0) + (8,), ("cnot2",) + cases.cnot2_case(qgd) + (8,),
                 ("cnot3",) + cases.cnot3_case(qgd, nsteps=12, tf=12.0) + (8,),
                 ("synthetic72",) + cases.synthetic_case(qgd, N=72, c=4, n_ops=2, nsteps=6, tf=0.3, seed=72) + (4,)]
        for name, prob, ctrl, pcof, target, order in todo:
            prob.gmres_abstol = prob.gmres_reltol = 1e-15
            g_ref, h_ref, lam_ref, f_ref, _ = orc.discrete_adjoint(prob, ctrl, pcof, target, order=order, return_all=True)
            lam = np.zeros(h_ref.shape, order="F")
            grad = np.zeros_like(g_ref)
            qgd.discrete_adjoint_(grad, None, lam, None, prob, ctrl, pcof, target, order=order, cost_type=":" + cost_type)
            # (:Norm on a closed system differentiates the scheme's loss of unitarity: the gradient is 1e-5 of the O(1)
            #  terms <lambda, dA w> it is summed from, so the bar is relative to those, not to the cancelled sum)
            scale = max(np.abs(g_ref).max(), 1e-3 * np.abs(lam_ref[:, 0]).max())
            assert np.abs(lam[:, 0] - lam_ref[:, 0]).max() <= 1e-10 * max(1.0, np.abs(lam_ref[:, 0]).max()), name
            assert np.abs(grad - g_ref).max() <= GRAD_RTOL * scale, name
            # the cost scalar and the stored forward sweep
            dp = qgd.device_problem(prob, order)
            dp.set_controls(ctrl); dp.set_target(target); dp.set_cost_type(cost_type)
            out3 = dp.eval_forward(pcof)
            wN, tr = h_ref[:, 0, -1, :], orc.target_real(target)
            cost = 0.5 * np.sum((wN - tr) ** 2) if cost_type == "Tracking" else 0.5 * np.sum(wN ** 2)
            assert abs(out3[0] - cost) <= 1e-11 * max(1.0, cost) and out3[1] == 0.0, name
            g_pre, _ = dp.discrete_adjoint(pcof, history_precomputed=True)
            assert np.abs(g_pre - g_ref).max() <= GRAD_RTOL * scale, name
            dp.set_cost_type("Infidelity")
            if prob.N_tot_levels <= 64:
                g_for = qgd.eval_grad_forced(prob, ctrl, pcof, target, order=order, cost_type=cost_type)
                assert np.abs(g_for - grad).max() <= 1e-11 * scale, name
                assert np.abs(g_for - orc.eval_grad_forced(prob, ctrl, pcof, target, order=order)).max() <= 1e-10 * scale, name
            if len(pcof) <= 50:
                g_fd = qgd.eval_grad_finite_difference(prob, ctrl, pcof, target, order=order, cost_type=cost_type)
                assert np.abs(g_fd - grad).max() <= 1e-8 * max(1.0, scale), name
            # the handle is back on :Infidelity afterwards
            orc.set_cost_type("Infidelity")
            g_inf = orc.discrete_adjoint(prob, ctrl, pcof, target, order=order)
            orc.set_cost_type(cost_type)
            assert np.abs(qgd.discrete_adjoint(prob, ctrl, pcof, target, order=order) - g_inf).max() <= GRAD_RTOL * np.abs(g_inf).max(), name
    finally:
        orc.set_cost_type("Infidelity")
        orc.set_converged_terminal(False)
    with pytest.raises(ValueError, match="Invalid cost type"):
        qgd.discrete_adjoint(prob, ctrl, pcof, target, order=order, cost_type="Fidelity")
    qgd.clear_cache()


def test_cost_type_stand_alone_terminal_kernel(qgd, orc, monkeypatch):
    """The same through k_terminal as its own launch (QGD_PATHS=terminal_kernel) instead of the extra workgroup of the first
    adjoint launch."""
    monkeypatch.setenv("QGD_PATHS", "terminal_kernel")
    prob, ctrl, pcof, target = cases.guarded_case(qgd, nsteps=20, tf=10.0)
    orc.set_converged_terminal(True); orc.set_cost_type("Tracking")
    try:
        g_ref = orc.discrete_adjoint(prob, ctrl, pcof, target, order=6)
    finally:
        orc.set_cost_type("Infidelity"); orc.set_converged_terminal(False)
    grad = qgd.discrete_adjoint(prob, ctrl, pcof, target, order=6, cost_type="Tracking")
    assert np.abs(grad - g_ref).max() <= GRAD_RTOL * np.abs(g_ref).max()
    qgd.clear_cache()


def test_randomised_shape_sweep_small_n():
    """scripts/fuzz_small_n.py: 40 random problems with N <= 64 (dispersive / sparse with guard levels and carrier
    controls, random dense; 1..24 columns, orders 2..16, 1..200 steps) against the numpy statement, history and
    gradient <= 1e-10 (its own process: every case creates and closes a handle)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "fuzz_small_n.py"), "40", "7"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "FAIL" not in r.stdout and "worst" in r.stdout


@pytest.mark.parametrize("which", ["cnot2", "cnot3"])
def test_baseline_configs_full_size_vs_oracle(qgd, orc, which):
    """BASELINE.json configs 2 and 3 at their FULL grids -- cnot2 (N=4, 4 columns, 100 steps) and cnot3 (N=64, 8 columns,
    550 steps at dt = 1), order 8 -- directly against the oracle: state history with all stage derivatives, guard forcing,
    lambda and the gradient.  The oracle runs its SparseMatrixCSC operator mode (bit-identical to its dense mode,
    test_oracle.py::test_csc_operators_equal_dense) on 8 threads over the columns, GMRES at 1e-15, terminal solve to
    convergence: 5.5 s for cnot3.  The history bound is the oracle's own: 550 steps of 1e-15 residuals through L^-1."""
    nsteps = 100 if which == "cnot2" else 550
    prob, ctrl, pcof, target = getattr(cases, which + "_case")(qgd, nsteps=nsteps, tf=float(nsteps))
    prob.gmres_abstol = prob.gmres_reltol = 1e-15
    orc.set_sparse_operators(True); orc.set_num_threads(8); orc.set_converged_terminal(True)
    try:
        g_ref, h_ref, lam_ref, f_ref, _ = orc.discrete_adjoint(prob, ctrl, pcof, target, order=8, return_all=True)
    finally:
        orc.set_sparse_operators(False); orc.set_converged_terminal(False); orc.set_num_threads(0)
    hist = np.zeros(h_ref.shape, order="F"); lam = np.zeros(h_ref.shape, order="F"); forcing = np.zeros(f_ref.shape, order="F")
    grad = np.zeros_like(g_ref)
    qgd.discrete_adjoint_(grad, hist, lam, forcing, prob, ctrl, pcof, target, order=8)
    assert np.abs(hist - h_ref).max() <= 5e-11
    assert np.abs(forcing - f_ref).max() <= 1e-12
    assert np.abs(lam[:, 0] - lam_ref[:, 0]).max() <= 1e-11 * np.abs(lam_ref[:, 0]).max()
    assert np.abs(grad - g_ref).max() <= GRAD_RTOL * np.abs(g_ref).max()
    qgd.clear_cache()


@pytest.mark.parametrize("order", [4, 8])
def test_nonlinear_controls_general_path(qgd, order):
    """Controls that are not linear in their coefficients (the AbstractControl protocol is open, Control.jl:6-27): the
    tables and their Jacobian at the current pcof come from the pointwise protocol (qgd_set_control_tables +
    qgd_set_control_basis, NULL pcof).  (1) A linear family seen through its pointwise protocol only reproduces the basis
    path: history bitwise-close, gradient to 1e-13.  (2) A sine control with amplitude, frequency and phase as
    coefficients: adjoint = forced on the device, both = centred differences, history_precomputed reuse."""
    prob, ctrl, pcof, target = cases.guarded_case(qgd, nsteps=24, tf=8.0)
    shape = (prob.real_system_size, 1 + order // 2, 1 + prob.nsteps, prob.N_initial_conditions)
    h_lin = np.zeros(shape, order="F"); h_gen = np.zeros(shape, order="F")
    g_lin = np.zeros(len(pcof)); g_gen = np.zeros(len(pcof))
    qgd.discrete_adjoint_(g_lin, h_lin, None, None, prob, ctrl, pcof, target, order=order)
    wrapped = [cases.PointwiseOnly(c) for c in ctrl]
    qgd.discrete_adjoint_(g_gen, h_gen, None, None, prob, wrapped, pcof, target, order=order)
    assert np.abs(h_gen - h_lin).max() <= 1e-13 * max(1.0, np.abs(h_lin).max())
    assert np.abs(g_gen - g_lin).max() <= 1e-13 * np.abs(g_lin).max()
    # (2)
    sines = [cases.SineControl(prob.tf), cases.SineControl(prob.tf)]
    th = np.array([0.08, 1.3, 0.4, 0.05, 0.7, -1.1])
    g_adj = qgd.discrete_adjoint(prob, sines, th, target, order=order)
    g_for = qgd.eval_grad_forced(prob, sines, th, target, order=order)
    g_fd = qgd.eval_grad_finite_difference(prob, sines, th, target, order=order, dpcof=1e-6)
    scale = np.abs(g_adj).max()
    assert np.abs(g_for - g_adj).max() <= 1e-11 * scale
    assert np.abs(g_fd - g_adj).max() <= 1e-7 * max(1.0, scale)
    dp = qgd.device_problem(prob, order)
    dp.set_controls(sines); dp.set_target(target)
    dp.eval_forward(th)
    g_pre, _ = dp.discrete_adjoint(th, history_precomputed=True)
    assert np.abs(g_pre - g_adj).max() <= 1e-13 * scale
    qgd.clear_cache()


def test_long_grid_vs_statement(qgd, orc):
    """cnot3 on a grid eight times the headline's (4400 steps at dt = 1: scan blocks of 69 steps, the inverse at 17
    matrices per CU) against the numpy statement of the algorithm: state history, infidelity, guard penalty, gradient."""
    nsteps = 4400
    prob, target = qgd.cnot3_problem(nsteps=nsteps, tf=float(nsteps))
    ctrl = cases.cnot3_controls(qgd, prob)
    npar = qgd.get_number_of_control_parameters(ctrl)
    pcof = (np.random.default_rng(3).random(npar) - 0.5) * 2 * np.pi * 0.005
    order = 8
    Gp, Gq, off = qgd.control_basis(ctrl, prob.nsteps, prob.tf, order // 2)
    ref = pp.evaluate(prob, Gp, Gq, off, pcof, target, order)
    cases.oracle_pins(orc, prob, ctrl, pcof, target, order, ref)
    dp = qgd.device_problem(prob, order)
    dp.set_controls(ctrl); dp.set_target(target)
    hist = np.zeros((128, 5, nsteps + 1, 8), order="F")
    grad, out3 = dp.discrete_adjoint(pcof, False, hist)
    href = pp.history_real(ref["ws"])
    assert np.abs(hist[:, 0] - href[:, 0]).max() <= 1e-10
    assert np.abs(hist - href).max() <= 1e-10 * max(1.0, np.abs(href).max())
    assert np.abs(grad - ref["grad"]).max() <= GRAD_RTOL * np.abs(ref["grad"]).max()
    infid = 1 - (out3[0] ** 2 + out3[1] ** 2) / prob.N_ess_levels ** 2
    assert abs(infid - ref["infidelity"]) <= 1e-11 and abs(out3[2] - ref["guard"]) <= 1e-11
    qgd.clear_cache()


@pytest.mark.parametrize("which,nsteps,order", [("cnot3", 550, 8), ("cnot3", 200, 4), ("guarded", 420, 6), ("cnot2", 300, 8), ("cnot3", 1100, 8)])
def test_adjoint_history_pass_suffix_products(qgd, which, nsteps, order, monkeypatch):
    """The adjoint history pass reaches the end of its block through the stored suffix product and affine part of the
    blocks behind it in the super-block (k_chain_fast MODE 6 beside the level-2 products, running values of the MODE 2
    level-2 chain) instead of stepping over them: same lambda and gradient as with QGD_PATHS=no_suffix (the stepping pass),
    on grids whose last super-block is full and not, and under a time partition."""
    import torch
    prob, ctrl, pcof, target = getattr(cases, which + "_case")(qgd, nsteps=nsteps, tf=float(nsteps) / 2)
    shape = (prob.real_system_size, 1 + order // 2, 1 + prob.nsteps, prob.N_initial_conditions)
    out = {}
    for off in (False, True):
        if off:
            monkeypatch.setenv("QGD_PATHS", "no_suffix")
        dp = qgd.DeviceProblem(prob, order); dp.set_controls(ctrl); dp.set_target(target)
        lam = np.zeros(shape, order="F")
        g, o = dp.discrete_adjoint(pcof, False, None, lam, None)
        out[off] = (g, np.asarray(o), lam)
        dp.close()
    assert np.abs(out[False][2] - out[True][2]).max() <= 1e-13 * max(1.0, np.abs(out[True][2]).max())
    assert np.abs(out[False][0] - out[True][0]).max() <= 1e-13 * np.abs(out[True][0]).max()
    monkeypatch.delenv("QGD_PATHS")
    stream = torch.cuda.current_stream().cuda_stream
    backs = [qgd.DeviceBackend(prob, order, ctrl, target, r, 2, device=0, stream=stream) for r in range(2)]
    for g, o in qgd.LocalGroup(backs).discrete_adjoint(pcof):
        assert np.abs(g - out[True][0]).max() <= 1e-12 * np.abs(out[True][0]).max()
    for b in backs:
        b.close()


@pytest.fixture(scope="module")
def inverse_bench_exe(tmp_path_factory):
    """scripts/ubench/inverse_cb_bench.hip compiled once for the module (hipcc is on the GPU box)."""
    import os, shutil, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path_factory.mktemp("ubench") / "inverse_cb_bench")
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", os.path.join(root, "quantumgatedesign.jl_amd", "csrc"),
                        "-I", os.path.join(root, "include"), os.path.join(root, "scripts", "ubench", "inverse_cb_bench.hip"), "-o", exe],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return exe


@pytest.mark.parametrize("nmat,data,first,later", [(300, 0, 0, 0), (300, 1, 1, 0), (300, 3, None, 0), (300, 4, 1, 1), (800, 0, 0, 0), (800, 1, 1, 0), (800, 4, 1, 1)])
def test_inverse_kernels_side_by_side(inverse_bench_exe, nmat, data, first, later):
    """Kernel level, outside the library's launch sequence: scripts/ubench/inverse_cb_bench.hip run on 300 (k_inverse_cb<true>:
    one round of workgroups) or 800 (k_inverse_cb<false>) random step matrices -- k_inverse_cb beside k_inverse_mfma<64> on the same L and R, `Linv L = I` and `L P = R` (both copies of
    P) checked on the host.  data 0: diagonally dominant (every matrix by the diagonal attempt); 1: rows permuted inside the
    16-row blocks (every matrix by the pivoted attempt); 3: noise 0.1 (some matrices leave the diagonal attempt); 4: entries of
    1000 outside the diagonal tiles (every matrix by the last resort).  first / later: the share of matrices past the first /
    second stage (None: some)."""
    import re, subprocess
    exe = inverse_bench_exe
    run = subprocess.run([exe, str(nmat), str(data)], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-2000:]
    errs = re.findall(r"(k_inverse_\w+)\S*\s+max \|Linv L - I\| = (\S+), max \|L P - R\| = (\S+) \(panel\) (\S+) \(planes\)", run.stdout)
    assert len(errs) == 2, run.stdout
    for name, e_inv, e_pp, e_pq in errs:
        assert float(e_inv) <= 1e-12 and float(e_pp) <= 2e-11 and float(e_pq) <= 2e-11, (name, e_inv, e_pp, e_pq)
    m = re.search(r"cb over all launches: (\d+) matrices not done by the diagonal attempt, (\d+) by the last resort", run.stdout)
    launches = 3 + 12 * 10      # warm-up + timed launches of the bench
    n_first, n_later = int(m.group(1)), int(m.group(2))
    if first is None:
        assert 0 < n_first < nmat * launches
    else:
        assert n_first == first * nmat * launches
    assert n_later == later * nmat * launches
    assert "status: mfma 0, cb 0" in run.stdout
