"""Seeded problem/control cases shared by the CPU and GPU parity tests.  The
shapes follow the reference's own tests (test/GradientTests/compare_gradients.jl:
Rabi 2-level and random N=4, GRAPE / B-spline degree 16 / carrier controls,
nsteps=10) plus the BASELINE.json configurations at reduced nsteps."""
import numpy as np


def rand_target(prob, seed=0):
    r = np.random.default_rng(seed)
    s = (prob.N_tot_levels, prob.N_initial_conditions)
    return r.random(s) + 1j * r.random(s)


def gradient_cases(qgd):
    """(name, prob, controls, pcof, target) as in compare_gradients.jl:103-230."""
    out = []
    rng = np.random.default_rng(0)
    mk = dict(gmres_abstol=1e-15, gmres_reltol=1e-15)

    def add(name, prob, ctrl):
        pcof = rng.random(qgd.get_number_of_control_parameters(ctrl))
        out.append((name, prob, ctrl, pcof, rand_target(prob)))

    p = qgd.construct_rabi_prob(tf=np.pi, nsteps=10, **mk)
    add("rabi-grape", p, qgd.GRAPEControl(5, p.tf))
    add("rabi-bspline16", p, qgd.FortranBSplineControl(16, 20, p.tf))
    add("rabi-carrier", p, qgd.CarrierControl(qgd.FortranBSplineControl(16, 20, p.tf), [-10, -1, 0, 1, 10]))
    p = qgd.construct_rand_prob(4, 1, tf=1.0, nsteps=10, **mk)
    add("rand4-grape", p, qgd.GRAPEControl(5, p.tf))
    p2 = qgd.construct_rand_prob(4, 1, tf=0.1, nsteps=10, **mk)
    add("rand4-bspline16", p2, qgd.FortranBSplineControl(16, 20, p2.tf))
    add("rand4-carrier", p, qgd.CarrierControl(qgd.FortranBSplineControl(16, 20, p.tf), [-10, -1, 0, 1, 10]))
    return out


def cnot2_case(qgd, nsteps=40, tf=40.0, seed=1, amp=0.05):
    prob, target = qgd.cnot2_problem(nsteps=nsteps, tf=tf)
    prob.gmres_abstol = prob.gmres_reltol = 1e-15
    ctrl = [qgd.GeneralBSplineControl(2, 10, prob.tf) for _ in range(prob.N_operators)]
    pcof = amp * (0.5 - np.random.default_rng(seed).random(qgd.get_number_of_control_parameters(ctrl)))
    return prob, ctrl, pcof, target


def guarded_case(qgd, nsteps=30, tf=15.0, seed=2):
    """Two 3-level qudits with a guard level each: exercises guard penalty/forcing, N=9 (padding)."""
    freqs = 2 * np.pi * np.array([4.1, 4.8])
    kerr = 2 * np.pi * np.array([[0.22, 0.01], [0.01, 0.225]])
    prob = qgd.DispersiveProblem((3, 3), (2, 2), freqs, freqs, kerr, tf, nsteps, gmres_abstol=1e-15, gmres_reltol=1e-15)
    target = qgd.create_gate((3, 3), (2, 2), [((1, 0), (1, 1)), ((1, 1), (1, 0))])
    ctrl = [qgd.CarrierControl(qgd.FortranBSplineControl(2, 8, prob.tf), [0.0, -kerr[0, 0]]) for _ in range(2)]
    pcof = 0.1 * (0.5 - np.random.default_rng(seed).random(qgd.get_number_of_control_parameters(ctrl)))
    return prob, ctrl, pcof, target


def cnot3_controls(qgd, prob):
    """BASELINE.md build choice: degree-2 B-spline, 10 basis functions per carrier, 3 carriers per control."""
    xa, xb = 2 * 0.1099, 2 * 0.1126
    carriers = [[0.0, -2 * np.pi * xa, -2 * np.pi * 1e-6], [0.0, -2 * np.pi * xb, -2 * np.pi * 1e-6],
                [0.0, -2 * np.pi * np.sqrt(xa * 0.002494 ** 2 / xa), -2 * np.pi * np.sqrt(xb * 0.002494 ** 2 / xa)]]
    return [qgd.CarrierControl(qgd.FortranBSplineControl(2, 10, prob.tf), carriers[k]) for k in range(3)]


def cnot3_case(qgd, nsteps=20, tf=20.0, seed=0):
    prob, target = qgd.cnot3_problem(nsteps=nsteps, tf=tf)
    prob.gmres_abstol = prob.gmres_reltol = 1e-15
    ctrl = cnot3_controls(qgd, prob)
    npar = qgd.get_number_of_control_parameters(ctrl)
    pcof = (np.random.default_rng(seed).random(npar) - 0.5) * 2 * np.pi * 0.005
    return prob, ctrl, pcof, target


def dense_guard_case(qgd, nsteps=16, tf=8.0, seed=4):
    """guarded_case with a dense symmetric (non-diagonal) guard matrix: "the projector actually
    doesn't need to be a projector" (SchrodingerProb.jl:137-141) -- exercises the general W kernel."""
    prob, ctrl, pcof, target = guarded_case(qgd, nsteps=nsteps, tf=tf, seed=seed)
    r = np.random.default_rng(seed).standard_normal((prob.real_system_size, prob.real_system_size))
    prob.guard_subspace_projector = np.asfortranarray(0.05 * (r + r.T))
    return prob, ctrl, pcof, target


def synthetic_case(qgd, N=100, c=20, n_ops=4, nsteps=24, tf=0.5, seed=5):
    """Shape of BASELINE.json configs[4] (random dense SchrodingerProb, entries scaled 1/N, degree-16
    B-spline controls with 20 basis functions, order 12) at a size the tests can afford; N=100 already
    takes every large-N fallback (HBM panel slabs, global-memory inverse, generic recursion kernels)."""
    prob = qgd.construct_rand_prob(N, n_ops, tf=tf, nsteps=nsteps, scale=1.0 / N)
    prob.u0 = np.asfortranarray(prob.u0[:, :c]); prob.v0 = np.asfortranarray(prob.v0[:, :c])
    prob.N_initial_conditions = c
    ctrl = [qgd.FortranBSplineControl(16, 20, prob.tf) for _ in range(n_ops)]
    rng = np.random.default_rng(seed)
    pcof = rng.random(qgd.get_number_of_control_parameters(ctrl))
    target = rng.random((N, c)) + 1j * rng.random((N, c))
    return prob, ctrl, pcof, target


class SineControl:
    """A control that is NOT linear in its coefficients theta = (A, omega, phi): p(t) = A sin(omega t + phi),
    q(t) = A/2 cos(omega t + phi) -- the reference's AbstractControl protocol is open (Control.jl:6-27), its in-tree
    families just happen to be linear.  Pointwise protocol only (unscaled time derivatives and their gradients)."""
    is_linear = False
    N_coeff = 3

    def __init__(self, tf):
        self.tf = float(tf)

    def _arg(self, t, th, d):
        return th[1] * t + th[2] + d * np.pi / 2

    def eval_p_derivative(self, t, th, d):
        return th[0] * th[1] ** d * np.sin(self._arg(t, th, d))

    def eval_q_derivative(self, t, th, d):
        return 0.5 * th[0] * th[1] ** d * np.cos(self._arg(t, th, d))

    def eval_grad_p_derivative(self, t, th, d):
        A, w, _ = th
        s, c = np.sin(self._arg(t, th, d)), np.cos(self._arg(t, th, d))
        return np.array([w ** d * s, A * (d * w ** (d - 1) if d else 0.0) * s + A * w ** d * t * c, A * w ** d * c])

    def eval_grad_q_derivative(self, t, th, d):
        A, w, _ = th
        s, c = np.sin(self._arg(t, th, d)), np.cos(self._arg(t, th, d))
        return 0.5 * np.array([w ** d * c, A * (d * w ** (d - 1) if d else 0.0) * c - A * w ** d * t * s, -A * w ** d * s])


class PointwiseOnly:
    """Any control seen through its pointwise protocol only (is_linear = False): sends a linear family down the
    general path, where it must reproduce the basis path."""
    is_linear = False

    def __init__(self, inner):
        self.inner, self.N_coeff, self.tf = inner, inner.N_coeff, inner.tf

    def eval_p_derivative(self, t, th, d): return self.inner.eval_p_derivative(t, th, d)
    def eval_q_derivative(self, t, th, d): return self.inner.eval_q_derivative(t, th, d)
    def eval_grad_p_derivative(self, t, th, d): return self.inner.eval_grad_p_derivative(t, th, d)
    def eval_grad_q_derivative(self, t, th, d): return self.inner.eval_grad_q_derivative(t, th, d)


def oracle_pins(orc, prob, ctrl, pcof, target, order, ref, tol=1e-10, budget=1.5e8):
    """Ties a result of tests/proto_propagator.py (the numpy statement of the DEVICE algorithm, which a test compares the device
    with at 1e-11 .. 1e-13) to the ORACLE (the reference's algorithm: GMRES per step, eval_grad_discrete_adjoint.jl) on the SAME
    inputs, inside the same test: gradient and state history with its stage derivatives at `tol`.  Skipped (returns False) when
    the oracle would take more than about ten seconds (N^2 * columns * steps * order^2 above `budget`)."""
    import numpy as np
    import proto_propagator as pp
    N, c = prob.N_tot_levels, prob.N_initial_conditions
    if float(N) * N * c * prob.nsteps * order * order > budget:
        return False
    import os
    orc.set_num_threads(min(8, os.cpu_count() or 1)); orc.set_converged_terminal(True)
    tols = (prob.gmres_abstol, prob.gmres_reltol)
    prob.gmres_abstol = prob.gmres_reltol = 1e-15      # (the oracle's step solves run to convergence: the device inverts L_n directly)
    try:
        g, h = orc.discrete_adjoint(prob, ctrl, pcof, target, order=order, return_all=True)[:2]
    finally:
        orc.set_converged_terminal(False); orc.set_num_threads(0)      # (0: the default, one thread per column up to the cores)
        prob.gmres_abstol, prob.gmres_reltol = tols
    assert np.abs(ref["grad"] - g).max() <= tol * np.abs(g).max(), "numpy statement vs oracle: gradient"
    # States at 1e-11.  Their stage derivatives w_j = (1/j) sum_i A_{j-1-i} w_i are a fixed linear map of the state of the
    # same time point whose norm grows like |A dt|^j / j!: a 2e-13 difference of two correctly rounded states is 1e-10 in w_6
    # of cnot3 at dt = 1 (measured, with the gradients of the two agreeing to 3e-15) -- hence the looser bound there.
    hs = pp.history_real(ref["ws"])
    assert np.abs(hs[:, 0] - h[:, 0]).max() <= 0.1 * tol * max(1.0, np.abs(h[:, 0]).max()), "numpy statement vs oracle: states"
    assert np.abs(hs - h).max() <= 10 * tol * max(1.0, np.abs(h).max()), "numpy statement vs oracle: stage derivatives"
    return True
