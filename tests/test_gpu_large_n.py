"""GPU tests of the N > 64 kernels: BASELINE.json configs[4] (C5) at its full shape, and a direct comparison of
the dense-GEMM path with the oracle (the reference-structured CPU restatement)."""
import numpy as np
import pytest

import cases
import proto_propagator as pp

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,c,n_ops,nsteps,order", [(80, 4, 2, 8, 12), (100, 3, 4, 6, 12), (72, 4, 1, 8, 4)])
def test_large_n_kernels_vs_oracle(qgd, orc, N, c, n_ops, nsteps, order):
    """The N > 64 path (fragment-ordered A_d/D_j GEMM kernels, blocked inverse, k_chain_dense) against the ORACLE --
    per-column matrix-free GMRES at 1e-15, exponential adjoint recursion, recursive gradient accumulation
    (hermite.jl, forward_evolution.jl, eval_grad_discrete_adjoint.jl) -- not against the numpy statement of the device
    algorithm: full derivative history, lambda, forcing and gradient."""
    prob, ctrl, pcof, target = cases.synthetic_case(qgd, N=N, c=c, n_ops=n_ops, nsteps=nsteps, tf=0.05 * nsteps, seed=N)
    prob.gmres_abstol = prob.gmres_reltol = 1e-15
    orc.set_converged_terminal(True)
    try:
        g_ref, h_ref, lam_ref, f_ref, _ = orc.discrete_adjoint(prob, ctrl, pcof, target, order=order, return_all=True)
    finally:
        orc.set_converged_terminal(False)
    hist = np.zeros(h_ref.shape, order="F"); lam = np.zeros(h_ref.shape, order="F")
    forcing = np.zeros(f_ref.shape, order="F")
    grad = np.zeros_like(g_ref)
    qgd.discrete_adjoint_(grad, hist, lam, forcing, prob, ctrl, pcof, target, order=order)
    for j in range(order // 2 + 1):      # relative per Taylor index: the high coefficients of a random problem are large
        assert np.abs(hist[:, j] - h_ref[:, j]).max() <= 1e-11 * max(1.0, np.abs(h_ref[:, j]).max()), j
    assert np.abs(lam[:, 0] - lam_ref[:, 0]).max() <= 1e-10 * max(1.0, np.abs(lam_ref[:, 0]).max())
    assert np.abs(forcing - f_ref).max() <= 1e-11 * max(1.0, np.abs(f_ref).max())
    assert np.abs(grad - g_ref).max() <= 1e-10 * np.abs(g_ref).max()
    qgd.clear_cache()


def test_config5_kernel_selection_vs_oracle(qgd, orc):
    """BASELINE.json configs[4] at the shape that SELECTS its kernels -- N = 256, 256 columns (the 16-column chain tiles, the
    outer-product gradient form "3" on N x N matrices: it needs 15 N < 19 c, block Gauss-Jordan over four 64-column blocks),
    4 control operators, order 12 -- against the ORACLE itself (per-column matrix-free GMRES at 1e-15, the exponential adjoint
    recursion and recursive_magic! of the reference), not the numpy statement of the device algorithm: 2 time steps,
    state history with all six stage derivatives, lambda, guard forcing and gradient.  The oracle's gradient loop runs
    its columns on threads for this one (bit-identical to the serial loop: tests/test_oracle.py); ~1 minute of host time.
    Reference shape: src/ProblemConstructors/random_problem.jl:15-35; contract: test/GradientTests/compare_gradients.jl:47-65."""
    import os
    N, c, n_ops, nsteps, order = 256, 256, 4, 2, 12
    prob, ctrl, pcof, target = cases.synthetic_case(qgd, N=N, c=c, n_ops=n_ops, nsteps=nsteps, tf=0.01 * nsteps, seed=N)
    prob.gmres_abstol = prob.gmres_reltol = 1e-15
    dp = qgd.DeviceProblem(prob, order); dp.set_controls(ctrl); dp.set_target(target)
    shape = (2 * N, 1 + order // 2, 1 + nsteps, c)
    hist, lam = np.zeros(shape, order="F"), np.zeros(shape, order="F")
    forcing = np.zeros((2 * N, 1 + nsteps, c), order="F")
    grad, _ = dp.discrete_adjoint(pcof, False, hist, lam, forcing)
    sel = dp.intermediate("selection")
    dp.close()
    assert list(sel) == [1.0, 3.0, 1.0, 1.0], sel      # GEMM-style kernels, gradient form 3, block inverse, one window
    orc.set_converged_terminal(True); orc.set_parallel_gradient(True)
    orc.set_num_threads(min(16, os.cpu_count() or 1))      # (the GPU box's CPU share for one GPU)
    try:
        g_ref, h_ref, lam_ref, f_ref, _ = orc.discrete_adjoint(prob, ctrl, pcof, target, order=order, return_all=True)
    finally:
        orc.set_converged_terminal(False); orc.set_parallel_gradient(False); orc.set_num_threads(0)
    for j in range(order // 2 + 1):      # relative per Taylor index: the high coefficients of a random problem are large
        assert np.abs(hist[:, j] - h_ref[:, j]).max() <= 1e-11 * max(1.0, np.abs(h_ref[:, j]).max()), j
    assert np.abs(lam[:, 0] - lam_ref[:, 0]).max() <= 1e-10 * max(1.0, np.abs(lam_ref[:, 0]).max())
    assert np.abs(forcing - f_ref).max() <= 1e-11 * max(1.0, np.abs(f_ref).max())
    assert np.abs(grad - g_ref).max() <= 1e-10 * np.abs(g_ref).max()


@pytest.mark.parametrize("N,c,n_ops,nsteps,order", [(80, 4, 2, 6, 12), (180, 9, 2, 4, 12)])
def test_large_n_lambda_derivative_columns(qgd, orc, N, c, n_ops, nsteps, order):
    """qgd_set_lambda_derivatives beyond N = 64 (the second case keeps the m+1 work panels of k_adjoint_derivs in HBM:
    7 x 192 x 16 doubles do not fit in LDS) against the oracle's tree recursion (hermite.jl:225-305)."""
    prob, ctrl, pcof, target = cases.synthetic_case(qgd, N=N, c=c, n_ops=n_ops, nsteps=nsteps, tf=0.05 * nsteps, seed=N)
    prob.gmres_abstol = prob.gmres_reltol = 1e-15
    orc.set_converged_terminal(True)
    try:
        g_ref, h_ref, lam_ref, f_ref, _ = orc.discrete_adjoint(prob, ctrl, pcof, target, order=order, return_all=True)
    finally:
        orc.set_converged_terminal(False)
    lam = np.zeros(h_ref.shape, order="F")
    grad = np.zeros_like(g_ref)
    qgd.discrete_adjoint_(grad, None, lam, None, prob, ctrl, pcof, target, order=order, lambda_derivatives=True)
    for j in range(order // 2 + 1):
        assert np.abs(lam[:, j] - lam_ref[:, j]).max() <= 1e-10 * max(1.0, np.abs(lam_ref[:, j]).max()), j
    assert np.abs(grad - g_ref).max() <= 1e-10 * np.abs(g_ref).max()
    qgd.clear_cache()


def c5_problem(qgd, nsteps, tf):
    """BASELINE.json configs[4] / SURVEY 8(d) C5: N=256=(4,4,4,4), 256 columns (U0 random complex N x N, N_ess = N),
    4 control operators, dense rand+rand^T / rand-rand^T entries scaled 1/N, degree-16 B-splines with 20 basis
    functions (shape of src/ProblemConstructors/random_problem.jl:15-35)."""
    prob, ctrl, pcof, _ = cases.synthetic_case(qgd, N=256, c=256, n_ops=4, nsteps=nsteps, tf=tf)
    return prob, ctrl, pcof, prob.u0 + 1j * prob.v0


def test_config5_reduced_steps_vs_statement(qgd):
    """C5's shape (N=256, 256 columns, 4 operators, order 12: the <2,4> 32-column chain tiles at full grid width,
    32 x 8 tiles) at 24 steps of the C5 step size: history (all Taylor indices) and gradient against the numpy
    statement of the algorithm, which tests/test_oracle.py ties to the oracle."""
    prob, ctrl, pcof, target = c5_problem(qgd, 24, 0.24)
    order = 12
    Gp, Gq, off = qgd.control_basis(ctrl, prob.nsteps, prob.tf, order // 2)
    ref = pp.evaluate(prob, Gp, Gq, off, pcof, target, order)
    dp = qgd.DeviceProblem(prob, order)
    dp.set_controls(ctrl); dp.set_target(target)
    hist = dp.pin(np.zeros((512, order // 2 + 1, prob.nsteps + 1, 256), order="F"))
    grad, out3 = dp.discrete_adjoint(pcof, False, hist)
    href = pp.history_real(ref["ws"])
    for j in range(order // 2 + 1):
        assert np.abs(hist[:, j] - href[:, j]).max() <= 1e-12 * max(1.0, np.abs(href[:, j]).max()), j
    assert np.abs(grad - ref["grad"]).max() <= 1e-10 * np.abs(ref["grad"]).max()
    del hist
    dp.close()


def test_config5_full_size_properties(qgd):
    """C5 at FULL size (N=256, 256 columns, 4 operators, order 12, tf=2, nsteps=200: the full block / super-block scan
    geometry), through size-independent properties:
      (1) the gradient equals a centred-difference directional derivative of infidelity + guard penalty;
      (2) history_precomputed reuse returns the same gradient and scalars;
      (3) the flow is unitary: the Gram matrix psi_n^H psi_n of the 256 columns is conserved over all 200 steps
          (order-12 steps at dt ||H|| ~ 0.03: to rounding);
      (4) the stored stage derivatives are consistent with the states: w_1(t_n) = A(t_n) w_0(t_n) through the
          Hamiltonian-application hook (hermite.jl:56-101, j = 0);
      (5) time windows: two in-process ranks reproduce the single-GPU gradient."""
    import torch
    prob, ctrl, pcof, target = c5_problem(qgd, 200, 2.0)
    order, N, c = 12, 256, 256
    dp = qgd.DeviceProblem(prob, order)
    dp.set_controls(ctrl); dp.set_target(target)
    hist = dp.pin(np.zeros((2 * N, order // 2 + 1, prob.nsteps + 1, c), order="F"))
    grad, out3 = dp.discrete_adjoint(pcof, False, hist)
    assert np.isfinite(grad).all() and np.isfinite(out3).all()
    # (2)
    grad_b, out3_b = dp.discrete_adjoint(pcof, history_precomputed=True)
    assert np.abs(grad_b - grad).max() <= 1e-12 * np.abs(grad).max()
    assert np.allclose(out3, out3_b, rtol=1e-13, atol=0)
    # (1)   the objective is ~ -3e4 here (random un-normalised U0 as target): rounding noise 1e-11 / eps
    d = np.random.default_rng(1).standard_normal(len(pcof)); d /= np.linalg.norm(d)
    eps = 1e-3

    def obj(p):
        a, b, g = dp.eval_forward(p)
        return 1 - (a * a + b * b) / prob.N_ess_levels ** 2 + g

    fd = (obj(pcof + eps * d) - obj(pcof - eps * d)) / (2 * eps)
    assert abs(fd - grad @ d) <= 1e-6 * abs(fd), (fd, grad @ d)
    # (3)
    psi0 = prob.u0 + 1j * prob.v0
    gram0 = psi0.conj().T @ psi0
    for n in (1, 67, 133, 200):
        psi = hist[:N, 0, n, :] + 1j * hist[N:, 0, n, :]
        assert np.abs(psi.conj().T @ psi - gram0).max() <= 1e-11 * np.abs(gram0).max(), n
    # (4)
    dp.eval_forward(pcof)          # tables of pcof on the device again (the FD evaluations changed them)
    for n in (0, 99, 200):
        w0 = np.asfortranarray(hist[:, 0, n, :])
        w1 = dp.apply_hamiltonian(w0, time_index=n, derivative_order=0)
        assert np.abs(w1 - hist[:, 1, n, :]).max() <= 1e-12 * np.abs(w1).max(), n
    del hist
    dp.close()
    # (5)
    stream = torch.cuda.current_stream().cuda_stream
    backs = [qgd.DeviceBackend(prob, order, ctrl, target, r, 2, device=0, stream=stream) for r in range(2)]
    for g, o in qgd.LocalGroup(backs).discrete_adjoint(pcof):
        assert np.abs(g - grad).max() <= 1e-11 * np.abs(grad).max()
        assert np.allclose(o, out3, rtol=1e-12, atol=0)
    for b in backs:
        b.close()


@pytest.mark.parametrize("N,c,n_ops,nsteps,order", [(72, 4, 1, 8, 4), (80, 3, 2, 7, 12), (100, 9, 2, 6, 8)])
def test_large_n_forced_gradient_vs_oracle(qgd, orc, N, c, n_ops, nsteps, order):
    """eval_grad_forced for N > 64 (src/eval_grad_forced.jl:17-194 has no size limit): the reference's own parity
    contract -- adjoint == forced (compare_gradients.jl:47-65) -- on the device, and the device's forced gradient
    against the ORACLE's eval_grad_forced.  (80, order 12) and (100, order 8) keep the work panels of k_forced_basis
    in HBM (more than 150 KB); the sensitivity scan runs in k_chain_forced_generic."""
    prob, ctrl, pcof, target = cases.synthetic_case(qgd, N=N, c=c, n_ops=n_ops, nsteps=nsteps, tf=0.05 * nsteps, seed=N)
    prob.gmres_abstol = prob.gmres_reltol = 1e-15
    prob.guard_subspace_projector = np.asfortranarray(np.diag(np.random.default_rng(N).random(2 * N)))      # exercise the guard part too
    g_orc = orc.eval_grad_forced(prob, ctrl, pcof, target, order=order)
    g_forced = qgd.eval_grad_forced(prob, ctrl, pcof, target, order=order)
    g_adj = qgd.discrete_adjoint(prob, ctrl, pcof, target, order=order)
    scale = np.abs(g_orc).max()
    assert np.abs(g_forced - g_orc).max() <= 1e-10 * scale
    assert np.abs(g_forced - g_adj).max() <= 1e-12 * scale
    qgd.clear_cache()


def test_config5_shape_adjoint_equals_forced(qgd):
    """C5's shape (N = 256, 256 columns, 4 control operators x 40 coefficients, order 12) at 24 steps: the discrete
    adjoint and the forward-sensitivity gradient agree to 1e-12 -- the reference's 1e-14-class cross-check
    (compare_gradients.jl:47-65) on a C5-shaped problem, where the oracle would need hours."""
    prob, ctrl, pcof, target = c5_problem(qgd, 24, 0.24)
    order = 12
    dp = qgd.DeviceProblem(prob, order)
    dp.set_controls(ctrl); dp.set_target(target)
    g_adj, _ = dp.discrete_adjoint(pcof)
    g_forced = dp.eval_grad_forced(pcof)
    dp.close()
    assert np.abs(g_forced - g_adj).max() <= 1e-12 * np.abs(g_adj).max()


@pytest.mark.parametrize("N,c,nsteps,order", [(72, 4, 8, 4), (100, 9, 6, 12)])
def test_large_n_forward_with_forcing_vs_oracle(qgd, orc, N, c, nsteps, order):
    """eval_forward(...; forcing) for N > 64 (forward_evolution.jl:118-129,167-206) against the oracle's forced sweep."""
    prob, ctrl, pcof, target = cases.synthetic_case(qgd, N=N, c=c, n_ops=2, nsteps=nsteps, tf=0.05 * nsteps, seed=N)
    prob.gmres_abstol = prob.gmres_reltol = 1e-15
    m = order // 2
    forcing = np.asfortranarray(0.3 * np.random.default_rng(3).standard_normal((2 * N, m, nsteps + 1, c)))
    h_ref = orc.eval_forward(prob, ctrl, pcof, order=order, forcing=forcing)
    hist = np.zeros(h_ref.shape, order="F")
    qgd.eval_forward_(hist, prob, ctrl, pcof, order=order, forcing=forcing)
    for j in range(m + 1):
        assert np.abs(hist[:, j] - h_ref[:, j]).max() <= 1e-11 * max(1.0, np.abs(h_ref[:, j]).max()), j
    qgd.clear_cache()


@pytest.mark.parametrize("N,c,n_ops,order", [(80, 16, 2, 12), (100, 40, 4, 8), (144, 144, 3, 4), (160, 150, 2, 12), (256, 64, 4, 12)])
def test_sigma_forms_agree(qgd, N, c, n_ops, order, monkeypatch):
    """The gradient scalars of the N > 64 path in their four forms -- operator applications (k_ginner_f: n_ops m GEMM
    units per time point), outer products over the columns with the stage derivatives (k_ginner_m: m(m+1)/2 units),
    outer products through the stored D_i (k_gouter + k_ginner_d: no stage derivatives), and the reverse sweep itself on
    the matrices Y_j = g_j psi_0^H (k_youter + k_yinit + k_gsweep_f + k_ginner_d) -- the last two only where their panels
    are allocated ((m-1) N < (m+1) c and N >= 128; else the request falls back to the second form) -- forced each way,
    on shapes with partial row blocks too (N = 80, 100: 5 and 7 blocks of 16), against the numpy statement."""
    prob, ctrl, pcof, target = cases.synthetic_case(qgd, N=N, c=c, n_ops=n_ops, nsteps=5, tf=0.05, seed=N)
    Gp, Gq, off = qgd.control_basis(ctrl, prob.nsteps, prob.tf, order // 2)
    ref = pp.evaluate(prob, Gp, Gq, off, pcof, target, order)
    grads = {}
    for f in ("0", "1", "2", "3"):
        monkeypatch.setenv("QGD_PATHS", "ginner=" + f)
        dp = qgd.DeviceProblem(prob, order); dp.set_controls(ctrl); dp.set_target(target)
        grads[f], _ = dp.discrete_adjoint(pcof)
        dp.close()
        assert np.abs(grads[f] - ref["grad"]).max() <= 1e-10 * np.abs(ref["grad"]).max(), f
    for f in ("1", "2", "3"):
        assert np.abs(grads["0"] - grads[f]).max() <= 1e-12 * np.abs(ref["grad"]).max(), f


@pytest.mark.parametrize("N,c,n_ops,order,form", [(128, 128, 4, 12, "3"), (128, 128, 2, 8, "2"), (144, 64, 4, 8, "1"),
                                                  (100, 32, 2, 8, "0"), (80, 24, 1, 4, "0")])
def test_three_product_tiles_against_four(qgd, N, c, n_ops, order, form, monkeypatch):
    """Since round 3 the N > 64 kernels form a complex product from THREE real ones (cgemm3_tile, outer_tile3, outer_frag3,
    k_chain_dense3: Re = P1 - P2, Im = P3 - P1 - P2).  The four-product kernels stay in the library (QGD_PATHS=dense_4m): both
    against the numpy statement (1e-10) and against each other (history 1e-12, gradient 1e-11 -- 3M is exact to
    eps |A||B| norm-wise, not component-wise), on shapes with an odd number of column groups (c = 24) and partial tiles."""
    prob, ctrl, pcof, target = cases.synthetic_case(qgd, N=N, c=c, n_ops=n_ops, nsteps=12, tf=0.12, seed=N + c)
    Gp, Gq, off = qgd.control_basis(ctrl, prob.nsteps, prob.tf, order // 2)
    ref = pp.evaluate(prob, Gp, Gq, off, pcof, target, order)
    out = {}
    for four in (False, True):
        monkeypatch.setenv("QGD_PATHS", "ginner=" + form + (",dense_4m" if four else ""))
        dp = qgd.DeviceProblem(prob, order); dp.set_controls(ctrl); dp.set_target(target)
        hist = np.zeros(dp._hist_shape(), order="F")
        g, o = dp.discrete_adjoint(pcof, False, hist)
        dp.close()
        assert np.abs(g - ref["grad"]).max() <= 1e-10 * np.abs(ref["grad"]).max(), four
        out[four] = (g, np.asarray(o), hist)
    assert np.abs(out[False][2] - out[True][2]).max() <= 1e-12 * np.abs(out[True][2]).max()
    assert np.abs(out[False][0] - out[True][0]).max() <= 1e-11 * np.abs(out[True][0]).max()
    assert np.abs(out[False][1] - out[True][1]).max() <= 1e-12 * max(1.0, np.abs(out[True][1]).max())


@pytest.mark.parametrize("N,c,order", [(256, 32, 8), (144, 24, 8), (80, 16, 4), (272, 16, 6)])
def test_block_gauss_jordan_inverse_and_its_fallback(qgd, N, c, order, monkeypatch):
    """Round 3: for N > 64 the step matrices are inverted by block Gauss-Jordan over 64-column blocks -- pivoting inside
    the diagonal blocks only, the block operations as batched GEMM tiles (qgdk_dense_inverse) -- and a matrix whose block
    multipliers exceed a threshold (or whose diagonal block meets a zero pivot) is redone by k_inverse_blocked2 with
    partial pivoting over whole columns.  Three routes must agree: the default (nothing redone on these matrices), every
    matrix forced through the fallback (QGD_PATHS=binv_thresh=0), and the round-2 kernel alone (QGD_PATHS=binv_off, a new handle);
    N = 80, 144, 272: last blocks of 16 columns."""
    prob, ctrl, pcof, target = cases.synthetic_case(qgd, N=N, c=c, n_ops=2, nsteps=6, tf=0.06, seed=3 * N)
    out = {}
    for tag, paths in (("default", ""), ("forced_fallback", "binv_thresh=0"), ("off", "binv_off")):
        monkeypatch.setenv("QGD_PATHS", paths)
        dp = qgd.DeviceProblem(prob, order); dp.set_controls(ctrl); dp.set_target(target)
        g, o = dp.discrete_adjoint(pcof)
        out[tag] = (g, np.asarray(o), dp.intermediate("Linv"), dp.intermediate("L"), int(dp.intermediate("repivoted")))
        dp.close()
    Linv, L = out["default"][2], out["default"][3]
    eye = np.eye(N)
    for n in range(1, prob.nsteps + 1):
        assert np.abs(Linv[n][:N, :N] @ L[n][:N, :N] - eye).max() <= 1e-12
    assert out["default"][4] == 0 and out["off"][4] == 0
    assert out["forced_fallback"][4] == prob.nsteps                 # every matrix of the grid went through the fallback
    for tag in ("forced_fallback", "off"):
        assert np.abs(out[tag][2] - Linv).max() <= 1e-12 * np.abs(Linv).max(), tag
        assert np.abs(out[tag][0] - out["default"][0]).max() <= 1e-11 * np.abs(out["default"][0]).max(), tag


def test_block_gauss_jordan_falls_back_on_large_time_steps(qgd, monkeypatch):
    """With time steps far beyond what the scheme is accurate at (dt |H| ~ 30: scripts/explore_binv.py prints the sweep) the
    leading blocks of L = sum_j c_j (-dt)^j D_j lose their conditioning (cond(L_11) ~ 1e4), some block multipliers pass the
    threshold, those matrices are redone with partial pivoting, and L^-1 L = I holds for every time point either way."""
    N = 128
    prob, ctrl, pcof, target = cases.synthetic_case(qgd, N=N, c=16, n_ops=2, nsteps=8, tf=8 * 64.0, seed=77)
    dp = qgd.DeviceProblem(prob, 4); dp.set_controls(ctrl); dp.set_target(target)
    dp.discrete_adjoint(pcof)
    Linv, L, redone = dp.intermediate("Linv"), dp.intermediate("L"), int(dp.intermediate("repivoted"))
    dp.close()
    assert redone > 0, "the case was meant to trip the multiplier check"
    for n in range(1, prob.nsteps + 1):
        r = np.abs(Linv[n][:N, :N] @ L[n][:N, :N] - np.eye(N)).max()
        assert r <= 1e-9 * max(1.0, np.linalg.cond(L[n][:N, :N])), (n, r)


@pytest.mark.parametrize("N,c,world", [(128, 64, 1), (144, 24, 3), (80, 40, 2)])
def test_sequential_chains_as_full_chip_steps(qgd, N, c, world, monkeypatch):
    """The single sequential chains of the scan on big problems (states at the super-block starts, the prefix over the
    lower ranks' windows, and their adjoint counterparts) run one full-chip GEMM launch per step (k_chain_step: 16 x 16
    output tiles, the four waves of a workgroup split the contraction) instead of one chain kernel on c/8 workgroups.
    The switch depends on the size (N^2 c >= 256^2 64); here the step path is forced on small shapes (QGD_PATHS=chain_steps_min=1)
    -- partial column pairs (c = 24, 40: 3 and 5 groups), N not a multiple of 64, a time partition of 2 and 3 ranks --
    against the chain kernels (chain_steps_min out of reach) and the numpy statement."""
    import torch
    order = 8
    prob, ctrl, pcof, target = cases.synthetic_case(qgd, N=N, c=c, n_ops=2, nsteps=40, tf=0.4, seed=N + world)
    Gp, Gq, off = qgd.control_basis(ctrl, prob.nsteps, prob.tf, order // 2)
    ref = pp.evaluate(prob, Gp, Gq, off, pcof, target, order)
    stream = torch.cuda.current_stream().cuda_stream
    res = {}
    for tag, paths in (("steps", "chain_steps_min=1"), ("chains", "chain_steps_min=1000000000000000")):
        monkeypatch.setenv("QGD_PATHS", paths)
        if world == 1:
            dp = qgd.DeviceProblem(prob, order); dp.set_controls(ctrl); dp.set_target(target)
            dp.set_timing(1)
            g, o = dp.discrete_adjoint(pcof)
            dp.close()
        else:
            backs = [qgd.DeviceBackend(prob, order, ctrl, target, r, world, device=0, stream=stream) for r in range(world)]
            g, o = qgd.LocalGroup(backs).discrete_adjoint(pcof)[world - 1]
            for b in backs:
                b.close()
        assert np.abs(g - ref["grad"]).max() <= 1e-10 * np.abs(ref["grad"]).max(), tag
        res[tag] = (g, np.asarray(o))
    assert np.abs(res["steps"][0] - res["chains"][0]).max() <= 1e-11 * np.abs(res["chains"][0]).max()
    assert np.abs(res["steps"][1] - res["chains"][1]).max() <= 1e-12 * max(1.0, np.abs(res["chains"][1]).max())


@pytest.mark.parametrize("N,c,n_ops,order,form", [(128, 128, 4, 8, "3"), (128, 128, 2, 8, "2"), (144, 64, 4, 8, "1"), (100, 32, 2, 8, "0"),
                                                  (80, 8, 1, 4, "0")])
def test_gradient_is_bitwise_reproducible_large_n(qgd, N, c, n_ops, order, form, monkeypatch):
    """The N > 64 gradient kernels add their scalars in a fixed order too (round 3: per-wave LDS slots, one plane of sigma per
    contributing tile, planes added in order by k_contract -- no atomicAdd left on the gradient path): the same inputs give the
    same bits, for every form of the gradient scalars, from one handle five times and from a fresh handle."""
    monkeypatch.setenv("QGD_PATHS", "ginner=" + form)
    prob, ctrl, pcof, target = cases.synthetic_case(qgd, N=N, c=c, n_ops=n_ops, nsteps=10, tf=0.1, seed=N)
    rng = np.random.default_rng(N)
    prob.guard_subspace_projector = np.asfortranarray(np.diag(rng.random(2 * N)))      # (a guard penalty to add up as well)
    grads, scalars = [], []
    for fresh in range(2):
        dp = qgd.DeviceProblem(prob, order); dp.set_controls(ctrl); dp.set_target(target)
        for _ in range(5 if not fresh else 1):
            g, o = dp.discrete_adjoint(pcof)
            grads.append(g); scalars.append(np.asarray(o))
        dp.close()
    assert scalars[0][2] > 0
    for g in grads[1:]:
        assert np.array_equal(g, grads[0])
    for o in scalars[1:]:
        assert np.array_equal(o, scalars[0])


@pytest.mark.parametrize("N,c,order,nsteps", [(300, 12, 4, 12), (320, 40, 8, 10), (512, 16, 6, 9), (592, 8, 4, 11)])
def test_sizes_beyond_288(qgd, N, c, order, nsteps):
    """Until the end of round 3 the GEMM-style chain kernels and the blocked inverse stopped at Np = 288 (their LDS), and a
    problem one size up fell onto the generic kernels: N = 288 took 72 ms where N = 256 took 5 (scripts/big_n_timing.py).
    The block Gauss-Jordan inverse has no such limit and the chains run on 16-column tiles up to Np = 640
    (k_chain_dense3<.,4,2>, four-product <.,5,2> beyond Np = 512): history and gradient against the numpy statement, and
    L^-1 L = I, at N = 300 (Np = 304), 320, 512 and 592."""
    prob, ctrl, pcof, target = cases.synthetic_case(qgd, N=N, c=c, n_ops=2, nsteps=nsteps, tf=0.01 * nsteps, seed=N)
    Gp, Gq, off = qgd.control_basis(ctrl, prob.nsteps, prob.tf, order // 2)
    ref = pp.evaluate(prob, Gp, Gq, off, pcof, target, order)
    dp = qgd.DeviceProblem(prob, order); dp.set_controls(ctrl); dp.set_target(target)
    hist = np.zeros(dp._hist_shape(), order="F")
    g, o = dp.discrete_adjoint(pcof, False, hist)
    Linv, L = dp.intermediate("Linv"), dp.intermediate("L")
    redone = int(dp.intermediate("repivoted"))
    dp.close()
    href = pp.history_real(ref["ws"])
    assert np.abs(hist - href).max() <= 1e-11 * np.abs(href).max()
    assert np.abs(g - ref["grad"]).max() <= 1e-10 * np.abs(ref["grad"]).max()
    assert redone == 0
    for n in (1, nsteps):
        assert np.abs(Linv[n][:N, :N] @ L[n][:N, :N] - np.eye(N)).max() <= 1e-12


def test_physical_four_qutrit_problem_vs_oracle(qgd, orc):
    """A problem of the reference's own kind beyond N = 64: four dispersively coupled qutrits (two essential + one guard
    level each: N = 81, 16 initial conditions, guard projector, rotating frame, B-splines with carrier waves) -- drift
    diagonal with entries of very different size, sparse couplings: the matrices the three-product tiles and the block
    Gauss-Jordan inverse were NOT tuned on.  Whole path against the oracle (matrix-free GMRES at 1e-15): history, lambda,
    guard forcing, gradient."""
    nsteps, order = 24, 8
    sizes, ess = (3, 3, 3, 3), (2, 2, 2, 2)
    freqs = 2 * np.pi * np.array([4.10, 4.35, 4.60, 4.85])
    kerr = 2 * np.pi * np.array([[0.20, 0.004, 0.003, 0.002], [0.004, 0.22, 0.005, 0.003], [0.003, 0.005, 0.21, 0.004], [0.002, 0.003, 0.004, 0.19]])
    prob = qgd.DispersiveProblem(sizes, ess, freqs, freqs, kerr, 0.5 * nsteps, nsteps, gmres_abstol=1e-15, gmres_reltol=1e-15)
    ctrl = [qgd.CarrierControl(qgd.FortranBSplineControl(2, 6, prob.tf), [0.0, -float(kerr[k, k])]) for k in range(prob.N_operators)]
    rng = np.random.default_rng(81)
    pcof = 0.05 * (rng.random(qgd.get_number_of_control_parameters(ctrl)) - 0.5)
    N, c = prob.N_tot_levels, prob.N_initial_conditions
    assert N == 81 and c == 16
    target = np.linalg.qr(rng.standard_normal((N, c)) + 1j * rng.standard_normal((N, c)))[0]
    orc.set_converged_terminal(True)
    try:
        g_ref, h_ref, lam_ref, f_ref, _ = orc.discrete_adjoint(prob, ctrl, pcof, target, order=order, return_all=True)
    finally:
        orc.set_converged_terminal(False)
    hist = np.zeros(h_ref.shape, order="F"); lam = np.zeros(h_ref.shape, order="F")
    forcing = np.zeros(f_ref.shape, order="F")
    grad = np.zeros_like(g_ref)
    qgd.discrete_adjoint_(grad, hist, lam, forcing, prob, ctrl, pcof, target, order=order)
    for j in range(order // 2 + 1):
        assert np.abs(hist[:, j] - h_ref[:, j]).max() <= 1e-11 * max(1.0, np.abs(h_ref[:, j]).max()), j
    assert np.abs(lam[:, 0] - lam_ref[:, 0]).max() <= 1e-10 * max(1.0, np.abs(lam_ref[:, 0]).max())
    assert np.abs(forcing - f_ref).max() <= 1e-11 * max(1.0, np.abs(f_ref).max())
    assert np.abs(grad - g_ref).max() <= 1e-10 * np.abs(g_ref).max()
    qgd.clear_cache()
