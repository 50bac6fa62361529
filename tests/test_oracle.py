"""CPU tests: the oracle against everything the reference's own tests pin for this path
(SURVEY.md section 8c).  No GPU needed."""
import math

import numpy as np
import pytest
import scipy.linalg

import cases
import proto_propagator as pp


def test_hermite_coefficients(orc):
    """c_j = m!(2m-j)!/((2m)!(m-j)!)  (hermite.jl:389-391); Pade(m,m) weights of exp."""
    assert orc.coefficient(0, 4, 4) == 1.0
    assert abs(orc.coefficient(1, 4, 4) - 0.5) < 1e-16
    assert abs(orc.coefficient(2, 4, 4) / 2 - 3 / 28) < 1e-16      # c_j/j! are the Pade weights
    assert abs(orc.coefficient(4, 4, 4) / 24 - 1 / 1680) < 1e-18
    assert abs(orc.coefficient(1, 1, 1) - 0.5) < 1e-16


def test_hardcoded_derivatives(qgd, orc):
    """Known-answer matrices of test/hardcoded_derivatives.jl:35-160: w', w'', w''', w''''
    and their transposes for A(t) = [0 K; -K 0], K = Ks + theta cos(t) Kc, to 1e-15."""
    rng = np.random.default_rng(7)
    w = rng.random(4); t = rng.random(); th = rng.random()
    Ks = np.array([[1.0, 0], [0, 0]]); Kc = np.array([[0.0, 1], [1, 0]]); Z = np.zeros((2, 2))
    blk = lambda K: np.block([[Z, K], [-K, Z]])
    A = blk(Ks + th * math.cos(t) * Kc)
    A1 = blk(-th * math.sin(t) * Kc); A2 = blk(-th * math.cos(t) * Kc); A3 = blk(th * math.sin(t) * Kc)
    M2 = A1 + A @ A
    M3 = A2 + 2 * A1 @ A + A @ A1 + A @ A @ A
    M4 = (A3 + 3 * A2 @ A + 3 * A1 @ A1 + 3 * A1 @ A @ A + A @ A2 + 2 * A @ A1 @ A + A @ A @ A1 + A @ A @ A @ A)
    hard = np.stack([w, A @ w, M2 @ w / 2, M3 @ w / 6, M4 @ w / 24], axis=1)
    hard_adj = np.stack([w, A.T @ w, M2.T @ w / 2, M3.T @ w / 6, M4.T @ w / 24], axis=1)
    prob = qgd.SchrodingerProb(Ks, Z, [Kc], [Z], np.zeros(2), np.zeros(2), None, 1.0, 1, 2)
    # SingleSymCosControl: p(t) = theta cos(t), q = 0 -> table rows p^(d)/d!
    pv = np.array([[th * math.cos(t)], [-th * math.sin(t)], [-th * math.cos(t) / 2], [th * math.sin(t) / 6],
                   [th * math.cos(t) / 24]])
    qv = np.zeros_like(pv)
    uv = np.zeros((4, 5), order="F"); uv[:, 0] = w
    got = orc.compute_derivatives(prob, pv, qv, uv)
    assert np.abs(got - hard).max() < 1e-15
    got_adj = orc.compute_derivatives(prob, pv, qv, uv, adjoint=True)
    assert np.abs(got_adj - hard_adj).max() < 1e-15


@pytest.mark.parametrize("order", [2, 4, 6, 8, 10, 12])
def test_constant_control_is_pade(qgd, orc, order):
    """With a constant Hamiltonian one step is the diagonal (m,m) Pade approximant of
    exp(A dt): the whole sweep vs scipy.linalg.expm."""
    prob = qgd.construct_rand_prob(4, 1, tf=0.5, nsteps=8, gmres_abstol=1e-15, gmres_reltol=1e-15)
    ctrl = qgd.GRAPEControl(1, prob.tf)
    pcof = np.array([0.3, -0.2])
    hist = orc.eval_forward(prob, ctrl, pcof, order=order)
    m = order // 2
    Hc = (prob.system_asym + pcof[1] * prob.asym_operators[0]) - 1j * (prob.system_sym + pcof[0] * prob.sym_operators[0])
    dt = prob.tf / prob.nsteps
    num = sum(orc.coefficient(j, m, m) * np.linalg.matrix_power(Hc * dt, j) / math.factorial(j) for j in range(m + 1))
    den = sum(orc.coefficient(j, m, m) * np.linalg.matrix_power(-Hc * dt, j) / math.factorial(j) for j in range(m + 1))
    step = np.linalg.solve(den, num)
    psi = prob.u0 + 1j * prob.v0
    for n in range(prob.nsteps):
        psi = step @ psi
    got = hist[:4, 0, -1, :] + 1j * hist[4:, 0, -1, :]
    assert np.abs(got - psi).max() < 1e-12
    if order >= 10:
        exact = scipy.linalg.expm(Hc * prob.tf) @ (prob.u0 + 1j * prob.v0)
        assert np.abs(got - exact).max() < 1e-9


def test_rabi_closed_form(qgd, orc):
    """H = [[0, p+iq],[p-iq, 0]]: U(t) = cos(|W|t) I - i sin(|W|t) H/|W|; |W|=1/2, tf=pi is a SWAP
    (rabi_oscillator.jl:1-22, optimization_rabi_osc_SWAP.jl:18-39)."""
    prob = qgd.construct_rabi_prob(tf=np.pi, nsteps=40, gmres_abstol=1e-15, gmres_reltol=1e-15)
    ctrl = qgd.GRAPEControl(1, prob.tf)
    for p, q_ in ((0.5, 0.0), (0.3, 0.4), (0.2, -0.1)):
        hist = orc.eval_forward(prob, ctrl, np.array([p, q_]), order=10)
        U = hist[:2, 0, -1, :] + 1j * hist[2:, 0, -1, :]
        W = math.hypot(p, q_)
        H = np.array([[0, p + 1j * q_], [p - 1j * q_, 0]])
        exact = math.cos(W * np.pi) * np.eye(2) - 1j * math.sin(W * np.pi) * H / W
        assert np.abs(U - exact).max() < 1e-11
    target = np.array([[0, -1j], [-1j, 0]])
    hist = orc.eval_forward(prob, ctrl, np.array([0.5, 0.0]), order=8)
    assert abs(orc.infidelity_real(hist[:, 0, -1, :], orc.target_real(target), 2)) < 1e-12


@pytest.mark.parametrize("order", [2, 4, 6, 8, 10])
def test_three_way_gradient_agreement(qgd, orc, order):
    """The reference's parity contract (test/GradientTests/compare_gradients.jl:47-65):
    discrete adjoint == forced to 1e-14, both == central differences (1e-5) to 1e-9."""
    for name, prob, ctrl, pcof, target in cases.gradient_cases(qgd):
        ga = orc.discrete_adjoint(prob, ctrl, pcof, target, order=order)
        gf = orc.eval_grad_forced(prob, ctrl, pcof, target, order=order)
        gd = orc.eval_grad_finite_difference(prob, ctrl, pcof, target, order=order, dpcof=1e-5)
        assert np.allclose(ga, gf, atol=1e-14, rtol=1e-14), (name, order, np.abs(ga - gf).max())
        # the reference uses 1e-9 on its MersenneTwister draws; ours differ, and the central
        # difference's own O(dpcof^2) truncation error reaches 1.05e-9 on one carrier case
        assert np.allclose(ga, gd, atol=3e-9, rtol=3e-9), (name, order, np.abs(ga - gd).max())
        assert np.allclose(gf, gd, atol=3e-9, rtol=3e-9), (name, order)


def test_three_way_with_guard_and_two_controls(qgd, orc):
    """Same contract on a problem with a guard projector and two carrier controls."""
    prob, ctrl, pcof, target = cases.guarded_case(qgd, nsteps=12, tf=6.0)
    for order in (4, 8):
        ga = orc.discrete_adjoint(prob, ctrl, pcof, target, order=order)
        gf = orc.eval_grad_forced(prob, ctrl, pcof, target, order=order)
        gd = orc.eval_grad_finite_difference(prob, ctrl, pcof, target, order=order, dpcof=1e-5)
        assert np.abs(ga - gf).max() < 1e-13
        assert np.abs(ga - gd).max() < 1e-8


@pytest.mark.parametrize("cost_type", ["Tracking", "Norm"])
def test_three_way_agreement_other_cost_types(qgd, orc, cost_type):
    """The same contract for cost_type = :Tracking / :Norm (eval_grad_discrete_adjoint.jl:26-35, eval_grad_forced.jl:
    160-163, eval_grad_finite_difference.jl:51-56), which the reference's own tests never exercise: the three gradients
    of the oracle agree, so its terminal right-hand sides are the derivatives of the costs the other two legs use."""
    orc.set_cost_type(cost_type)
    try:
        todo = [c + (o,) for c in cases.gradient_cases(qgd)[:4] for o in (2, 6)]
        todo.append(("guarded",) + cases.guarded_case(qgd, nsteps=10, tf=5.0) + (4,))
        for name, prob, ctrl, pcof, target, order in todo:
            ga = orc.discrete_adjoint(prob, ctrl, pcof, target, order=order)
            gf = orc.eval_grad_forced(prob, ctrl, pcof, target, order=order)
            gd = orc.eval_grad_finite_difference(prob, ctrl, pcof, target, order=order, dpcof=1e-5)
            scale = max(1.0, np.abs(gf).max())
            assert np.abs(ga - gf).max() <= 1e-13 * scale, (name, order)
            assert np.abs(ga - gd).max() <= 2e-8 * scale, (name, order)
    finally:
        orc.set_cost_type("Infidelity")
    # and the switch is really off again
    name, prob, ctrl, pcof, target = cases.gradient_cases(qgd)[0]
    assert np.abs(orc.discrete_adjoint(prob, ctrl, pcof, target, order=2) - orc.eval_grad_forced(prob, ctrl, pcof, target, order=2)).max() < 1e-14


@pytest.mark.parametrize("order", [2, 4, 6, 8, 10])
def test_convergence_order(qgd, orc, order):
    """Step doubling: observed order within +-0.5 of nominal
    (test/ConvergenceTests/forward_convergence.jl:47-65; Richardson, src/Tests/test_convergence.jl:238-250).
    Random N=4 problem, smooth (single-piece degree-16) B-spline control."""
    base = 16
    sols = []
    for nsteps in (base, 2 * base, 4 * base):
        prob = qgd.construct_rand_prob(4, 1, tf=1.0, nsteps=nsteps, gmres_abstol=1e-15, gmres_reltol=1e-15, scale=0.5)
        ctrl = qgd.FortranBSplineControl(16, 17, prob.tf)
        pcof = 0.5 * np.random.default_rng(3).random(ctrl.N_coeff)
        sols.append(orc.eval_forward(prob, ctrl, pcof, order=order)[:, 0, -1, :])
    # Richardson: ||u_h - u_{h/2}|| / ||u_{h/2} - u_{h/4}|| = 2^order
    obs = math.log2(np.linalg.norm(sols[0] - sols[1]) / np.linalg.norm(sols[1] - sols[2]))
    assert abs(obs - order) < 0.5, (order, obs)


@pytest.mark.parametrize("which,order", [("cnot2", 2), ("cnot2", 8), ("guarded", 6), ("guarded", 12)])
def test_propagator_form_equals_reference_form(qgd, orc, which, order):
    """The device algorithm (explicit step propagators, O(m^2) reverse sweep; numpy statement in
    proto_propagator.py) reproduces the reference algorithm (GMRES per step, exponential
    recursions) to rounding."""
    prob, ctrl, pcof, target = getattr(cases, which + "_case")(qgd)
    m = order // 2
    Gp, Gq, off = qgd.control_basis(ctrl, prob.nsteps, prob.tf, m)
    r = pp.evaluate(prob, Gp, Gq, off, pcof, target, order)
    g, hist, lam, forcing, _ = orc.discrete_adjoint(prob, ctrl, pcof, target, order=order, return_all=True)
    assert np.abs(pp.history_real(r["ws"]) - hist).max() < 1e-12
    assert np.abs(g - r["grad"]).max() <= 1e-12 * np.abs(g).max()
    assert abs(r["guard"] - orc.guard_penalty_real(prob, hist)) < 1e-13


def test_reference_terminal_solve_stalls(qgd, orc):
    """Documented behaviour of the reference on cnot3 at dt = 1: the one-shot gmres! of
    compute_terminal_condition runs with IterativeSolvers' defaults (restart 20, 2N iterations,
    eval_grad_discrete_adjoint.jl:61-62), needs ~90 Krylov vectors and stops short; the gradient
    is then off by ~1e-6 relative.  With the solve run to convergence the oracle agrees with the
    exact discrete gradient to rounding."""
    prob, ctrl, pcof, target = cases.cnot3_case(qgd, nsteps=6, tf=6.0)
    Gp, Gq, off = qgd.control_basis(ctrl, prob.nsteps, prob.tf, 4)
    exact = pp.evaluate(prob, Gp, Gq, off, pcof, target, 8)["grad"]
    g_faithful = orc.discrete_adjoint(prob, ctrl, pcof, target, order=8)
    orc.set_converged_terminal(True)
    try:
        g_conv = orc.discrete_adjoint(prob, ctrl, pcof, target, order=8)
    finally:
        orc.set_converged_terminal(False)
    rel = lambda g: np.abs(g - exact).max() / np.abs(exact).max()
    assert rel(g_conv) < 1e-12
    assert 1e-9 < rel(g_faithful) < 1e-3


def test_csc_operators_equal_dense(qgd, orc):
    """The oracle's SparseMatrixCSC operator mode (the reference's DispersiveProblem default sparse_rep=true,
    multi_qudit_systems.jl:118-162; used by bench.py's CPU-baseline variants) performs the same additions in the same
    order over the stored entries: identical histories and gradient."""
    prob, ctrl, pcof, target = cases.guarded_case(qgd, nsteps=6, tf=3.0)
    out = []
    for sp in (False, True):
        orc.set_sparse_operators(sp)
        try:
            out.append(orc.discrete_adjoint(prob, ctrl, pcof, target, order=6, return_all=True))
        finally:
            orc.set_sparse_operators(False)
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])


def test_hardcoded_partial_derivatives(qgd, orc):
    """The d/dtheta known-answer block of test/hardcoded_derivatives.jl:137-160 (compute_partial_derivative!, hermite.jl:
    321-386) against BOTH gradient routes of the oracle, to 1e-15: (a) the forced route -- the Taylor recursion on the
    partial-derivative matrix with the (dA/dtheta) w_i terms as its forcing (eval_grad_forced.jl:95-131); (b) the adjoint
    route -- recursive_magic! with compute_inner_prod_S!/K! (eval_grad_discrete_adjoint.jl:656-800), which must return
    coeff * <d w_k/d theta, lambda> for any lambda.  p(t) = theta cos(t) is CarrierControl(GRAPE(1), [1.0]) with
    pcof = [theta, 0] (its q = theta sin(t) multiplies the zero antisymmetric operator of the test)."""
    rng = np.random.default_rng(11)
    w = rng.random(4); t = rng.random(); th = rng.random()
    Ks = np.array([[1.0, 0], [0, 0]]); Kc = np.array([[0.0, 1], [1, 0]]); Z = np.zeros((2, 2))
    blk = lambda K: np.block([[Z, K], [-K, Z]])
    c, s = math.cos(t), math.sin(t)
    A = blk(Ks + th * c * Kc); A1 = blk(-th * s * Kc); A2 = blk(-th * c * Kc)
    dA = blk(c * Kc); dA1 = blk(-s * Kc); dA2 = blk(-c * Kc)
    hard = np.stack([np.zeros(4), dA @ w, (dA1 + dA @ A + A @ dA) @ w / 2,
                     (dA2 + 2 * dA1 @ A + 2 * A1 @ dA + dA @ A1 + A @ dA1 + dA @ A @ A + A @ dA @ A + A @ A @ dA) @ w / 6], axis=1)
    prob = qgd.SchrodingerProb(Ks, Z, [Kc], [Z], np.zeros(2), np.zeros(2), None, 1.0, 1, 2)
    ctrl = qgd.CarrierControl(qgd.GRAPEControl(1, 1.0), [1.0])
    pcof = np.array([th, 0.0])
    m = 3
    pv = np.array([orc.fill_p_vec(ctrl, t, pcof, m + 1)]).T          # [(1+m), 1]: p^(d)/d!
    qv = np.array([orc.fill_p_vec(ctrl, t, pcof, m + 1, q=True)]).T
    assert np.abs(pv[:, 0] - th * np.array([c, -s, -c / 2, s / 6])).max() < 1e-16
    uv = np.zeros((4, m + 1), order="F"); uv[:, 0] = w
    wm = orc.compute_derivatives(prob, pv, qv, uv)                   # w_0 .. w_3
    # (a) forcing of the partial-derivative recursion: F_j = sum_{i<=j} (dA_{j-i}/dtheta) w_i, dA_d/dtheta = (d^d/dt^d cos t / d!) [0 Kc; -Kc 0]
    gp = [orc.eval_grad_derivative(ctrl, t, pcof, d)[0] / math.factorial(d) for d in range(m)]
    assert np.abs(np.array(gp) - np.array([c, -s, -c / 2])).max() < 1e-16
    F = np.zeros((4, m), order="F")
    for j in range(m):
        for i in range(j + 1):
            F[:, j] += gp[j - i] * (blk(Kc) @ wm[:, i])
    part = orc.compute_derivatives(prob, pv, qv, np.zeros((4, m + 1), order="F"), forcing=F)
    assert np.abs(part - hard).max() < 1e-15
    # (b) recursive_magic! for k = 0..3
    lam = rng.standard_normal(4)
    for k in range(m + 1):
        coeff = 0.37 * (k + 1)
        got = orc.recursive_magic(prob, ctrl, pcof, m, 0, t, wm, lam, k, coeff)
        assert abs(got[0] - coeff * (hard[:, k] @ lam)) < 2e-15, (k, got, coeff * (hard[:, k] @ lam))


@pytest.mark.parametrize("which,order", [("cnot3", 8), ("cnot3", 12), ("synthetic72", 12), ("synthetic80", 4)])
def test_propagator_form_equals_reference_form_large(qgd, orc, which, order):
    """The same tie as above where half of the GPU suite leans on the numpy statement: Hermite order 12 and N >= 64
    (cnot3: N = 64 with guard levels; dense random N = 72 / 80 at order 12 / 4).  State history with every stage
    derivative, adjoint forcing, lambda and gradient of the statement against the reference-structured oracle (its
    terminal solve run to convergence, see test_reference_terminal_solve_stalls)."""
    if which == "cnot3":
        # (order 12 at dt = 0.25: at dt = 1 the oracle's own GMRES residuals accumulate to 1e-10 in the m = 6 stage derivatives)
        nsteps = 20 if order == 8 else 8
        prob, ctrl, pcof, target = cases.cnot3_case(qgd, nsteps=nsteps, tf=float(nsteps) * (1.0 if order == 8 else 0.25))
    else:
        N = int(which[len("synthetic"):])
        prob, ctrl, pcof, target = cases.synthetic_case(qgd, N=N, c=3, nsteps=5, tf=0.1)
        prob.gmres_abstol = prob.gmres_reltol = 1e-15
    m = order // 2
    Gp, Gq, off = qgd.control_basis(ctrl, prob.nsteps, prob.tf, m)
    r = pp.evaluate(prob, Gp, Gq, off, pcof, target, order)
    orc.set_converged_terminal(True)
    orc.set_num_threads(8)
    try:
        g, hist, lam, forcing, _ = orc.discrete_adjoint(prob, ctrl, pcof, target, order=order, return_all=True)
    finally:
        orc.set_converged_terminal(False)
    hs = max(1.0, np.abs(hist).max())
    assert np.abs(pp.history_real(r["ws"]) - hist).max() < 1e-11 * hs
    assert np.abs(g - r["grad"]).max() <= 1e-10 * np.abs(g).max()
    assert abs(r["guard"] - orc.guard_penalty_real(prob, hist)) < 1e-12


def test_legacy_controls_oracle_vs_host(qgd, orc):
    """the hard-coded quadratic B-spline (BSpline2Control, bspline_control.jl:21-249) and the Juqbox
    bcarrier2 layout (BSplineControl, bspline_control.jl:251-395 + bspline_backend.jl:381-955) restated in the oracle,
    against the product's host classes: values p^(d), q^(d) and the d/dtheta rows at knots, end points and random
    times.  BSplineControl is checked through the oracle's direct bcarrier2 restatement (orders 0, 1 -- all the
    reference implements), which is independent of the carrier-wave wrapper the product builds it from."""
    rng = np.random.default_rng(21)
    tf, D1 = 7.0, 9
    omega = [0.0, 1.3, -2.1]
    b2 = qgd.BSpline2Control(D1, tf)
    bc_host = qgd.BSplineControl(tf, D1, omega)
    from oracle.oracle import BCarrier2Control
    bc_orc = BCarrier2Control(tf, D1, omega)
    assert bc_host.N_coeff == bc_orc.N_coeff == 2 * D1 * len(omega)
    dtk = tf / (D1 - 2)
    times = np.concatenate([[0.0, tf, dtk, 3 * dtk, tf - 1e-13], rng.random(12) * tf])
    for host, oc, dmax in ((b2, b2, 3), (bc_host, bc_orc, 1)):
        pcof = rng.standard_normal(host.N_coeff)
        for t in times:
            for d in range(dmax + 1):
                for is_q in (False, True):
                    want = (host.eval_q_derivative if is_q else host.eval_p_derivative)(t, pcof, d)
                    got = orc.fill_p_vec(oc, t, pcof, d + 1, q=is_q)[d] * math.factorial(d)
                    assert abs(got - want) <= 1e-13 * max(1.0, abs(want)), (type(oc).__name__, t, d, is_q)
                    gw = (host.eval_grad_q_derivative if is_q else host.eval_grad_p_derivative)(t, pcof, d)
                    gg = orc.eval_grad_derivative(oc, t, pcof, d, q=is_q)
                    assert np.abs(gg - gw).max() <= 1e-13 * max(1.0, np.abs(gw).max()), (type(oc).__name__, t, d, is_q)
    # the spline is a partition of unity inside [0, tf] and its derivative rows sum to zero (bspline2's three pieces)
    ones = np.concatenate([np.ones(D1), np.zeros(D1)])
    for t in times:
        v = orc.fill_p_vec(b2, t, ones, 3)
        assert abs(v[0] - 1.0) < 1e-13 and abs(v[1]) < 1e-12 and abs(v[2]) < 1e-12


def test_legacy_controls_three_way_contract(qgd, orc):
    """The reference's parity contract (test/GradientTests/compare_gradients.jl:47-65: adjoint == forced <= 1e-14,
    == central differences <= 3e-9) met by the oracle with the legacy control families: BSpline2Control under carrier
    waves (my_bspline_controls, bspline_control.jl:52-64) at orders 2-6, and BSplineControl in its bcarrier2 form at
    order 2 (it implements derivative orders 0 and 1 only)."""
    from oracle.oracle import BCarrier2Control
    prob = qgd.construct_rand_prob(4, 1, tf=1.0, nsteps=10, gmres_abstol=1e-15, gmres_reltol=1e-15)
    target = cases.rand_target(prob)
    rng = np.random.default_rng(5)
    for ctrl, orders in ((qgd.CarrierControl(qgd.BSpline2Control(6, prob.tf), [-3.0, 0.0, 2.0]), (2, 4, 6)),
                         (BCarrier2Control(prob.tf, 6, [-3.0, 0.0, 2.0]), (2,))):
        pcof = rng.random(ctrl.N_coeff)
        for order in orders:
            ga = orc.discrete_adjoint(prob, ctrl, pcof, target, order=order)
            gf = orc.eval_grad_forced(prob, ctrl, pcof, target, order=order)
            gd = orc.eval_grad_finite_difference(prob, ctrl, pcof, target, order=order)
            scale = np.abs(ga).max()
            assert np.abs(ga - gf).max() <= 1e-13 * scale, (type(ctrl).__name__, order)
            assert np.abs(ga - gd).max() <= 3e-8 * scale, (type(ctrl).__name__, order)


def test_parallel_gradient_columns_are_bit_identical(qgd, orc):
    """The oracle's test switch for large problems (qo_set_parallel_gradient: the per-column gradient accumulation of
    eval_grad_discrete_adjoint.jl:148-157 with the columns on threads, each into its own vector, added in column order)
    gives the SAME BITS as the reference's serial loop -- so the N = 256 device comparison that uses it is a comparison
    with the serial oracle."""
    prob = qgd.construct_rand_prob(6, 2, tf=1.0, nsteps=8, gmres_abstol=1e-14, gmres_reltol=1e-14)
    ctrl = [qgd.FortranBSplineControl(3, 6, prob.tf) for _ in range(2)]
    rng = np.random.default_rng(3)
    pcof = rng.standard_normal(qgd.get_number_of_control_parameters(ctrl))
    target = cases.rand_target(prob)
    orc.set_num_threads(4)
    try:
        g_serial = orc.discrete_adjoint(prob, ctrl, pcof, target, order=6)
        orc.set_parallel_gradient(True)
        g_par = orc.discrete_adjoint(prob, ctrl, pcof, target, order=6)
    finally:
        orc.set_parallel_gradient(False)
        orc.set_num_threads(0)
    assert np.array_equal(g_serial, g_par) and np.abs(g_serial).max() > 0


@pytest.mark.parametrize("which,order", [("cnot2", 2), ("cnot2", 8), ("guarded", 6), ("cnot3", 8), ("dense_guard", 4)])
def test_local_front_form_equals_propagator_form(qgd, orc, which, order):
    """Round 6: the same-point propagator S_n = R_n L_n^-1 (forward sweep in phi = L psi, adjoint sweep directly in lambda:
    proto_propagator.evaluate_local, the statement of csrc/qgd_k_front.hip's evaluation order) gives the state history, lambda,
    guard and gradient of the two-point form P_n = L_{n+1}^-1 R_n to rounding, and those of the reference-structured oracle."""
    prob, ctrl, pcof, target = getattr(cases, which + "_case")(qgd)
    m = order // 2
    Gp, Gq, off = qgd.control_basis(ctrl, prob.nsteps, prob.tf, m)
    r = pp.evaluate(prob, Gp, Gq, off, pcof, target, order)
    q = pp.evaluate_local(prob, Gp, Gq, off, pcof, target, order)
    gs = np.abs(r["grad"]).max()
    assert np.abs(q["psi"] - r["psi"]).max() < 1e-13
    assert np.abs(q["lam"] - r["lam"]).max() <= 1e-12 * max(1e-30, np.abs(r["lam"]).max())
    assert np.abs(q["grad"] - r["grad"]).max() <= 1e-12 * gs
    assert abs(q["guard"] - r["guard"]) < 1e-14 and abs(q["infidelity"] - r["infidelity"]) < 1e-13
    assert cases.oracle_pins(orc, prob, ctrl, pcof, target, order, q)


def test_local_front_form_cnot3_headline_grid(qgd, orc):
    """Gate (i) of the round-6 plan: on the headline grid (cnot3, order 8, 550 steps of dt = 1) the local form agrees with the
    two-point form to 1e-12 and with the oracle (GMRES per step at 1e-15, converged terminal solve) to 1e-10 on the gradient."""
    prob, ctrl, pcof, target = cases.cnot3_case(qgd, nsteps=550, tf=550.0)
    Gp, Gq, off = qgd.control_basis(ctrl, prob.nsteps, prob.tf, 4)
    r = pp.evaluate(prob, Gp, Gq, off, pcof, target, 8)
    q = pp.evaluate_local(prob, Gp, Gq, off, pcof, target, 8)
    gs = np.abs(r["grad"]).max()
    assert np.abs(q["grad"] - r["grad"]).max() <= 1e-12 * gs
    assert np.abs(q["psi"] - r["psi"]).max() < 1e-12
    assert np.abs(q["lam"] - r["lam"]).max() <= 1e-12 * np.abs(r["lam"]).max()
    assert cases.oracle_pins(orc, prob, ctrl, pcof, target, 8, q, budget=1e12)
