"""Control functions p(t), q(t) -- host-side mirror of the reference's
``AbstractControl`` protocol (reference: src/Controls/Control.jl:6-27).

The stepper never calls a control object in its inner loop.  It consumes, per
time point, the table ``vals[d, k] = p_k^(d)(t)/d!`` that the reference builds
with ``fill_p_mat!/fill_q_mat!`` (Control.jl:125-149).  Every control family in
scope (GRAPE, clamped B-splines, carrier waves) is *linear* in its coefficients,
so the whole table is ``G @ pcof`` with a basis tensor ``G`` that depends only on
the time grid.  ``G`` is built here once (numpy, setup cost), uploaded once, and
the same tensor is the d(table)/d(pcof) the adjoint gradient needs.

Names and argument meaning follow the reference:
  GRAPEControl(N_amplitudes, tf)                  grape_control.jl:18-28
  FortranBSplineControl(degree, N_basis, tf)      FortranBSpline.jl:17-60
  GeneralBSplineControl(degree, N_knots, tf)      GeneralBSplineControl.jl:1-19
  CarrierControl(base_control, carrier_freqs)     CarrierControl.jl:5-24
"""
from __future__ import annotations

import math
from typing import Sequence

import numpy as np


class AbstractControl:
    """Protocol: ``N_coeff``, ``tf`` and the eval_* methods (Control.jl:6-27)."""

    N_coeff: int
    tf: float
    is_linear = True  # p, q are linear in pcof (true for every family below)

    # -- tables over a time grid (what the stepper consumes) ----------------
    def grad_tables(self, times: np.ndarray, dmax: int):
        """(Gp, Gq), each ``[len(times), dmax+1, N_coeff]``:
        d/dpcof_l of the *unscaled* d-th time derivative of p (q) at each time.
        Mirrors eval_grad_p_derivative!/eval_grad_q_derivative! over a grid."""
        raise NotImplementedError

    # -- pointwise protocol (reference signatures) --------------------------
    def eval_grad_p_derivative(self, t, pcof, order):
        return self.grad_tables(np.array([float(t)]), order)[0][0, order].copy()

    def eval_grad_q_derivative(self, t, pcof, order):
        return self.grad_tables(np.array([float(t)]), order)[1][0, order].copy()

    def eval_p_derivative(self, t, pcof, order):
        return float(self.eval_grad_p_derivative(t, pcof, order) @ np.asarray(pcof, float))

    def eval_q_derivative(self, t, pcof, order):
        return float(self.eval_grad_q_derivative(t, pcof, order) @ np.asarray(pcof, float))

    def eval_p(self, t, pcof):
        return self.eval_p_derivative(t, pcof, 0)

    def eval_q(self, t, pcof):
        return self.eval_q_derivative(t, pcof, 0)

    def fill_p_vec(self, vals, t, pcof):
        """vals[d] = p^(d)(t)/d!  (Control.jl:99-110)."""
        gp, _ = self.grad_tables(np.array([float(t)]), len(vals) - 1)
        for d in range(len(vals)):
            vals[d] = gp[0, d] @ np.asarray(pcof, float) / math.factorial(d)
        return vals

    def fill_q_vec(self, vals, t, pcof):
        _, gq = self.grad_tables(np.array([float(t)]), len(vals) - 1)
        for d in range(len(vals)):
            vals[d] = gq[0, d] @ np.asarray(pcof, float) / math.factorial(d)
        return vals

    # Control.jl:42-58 -- a single control behaves like a 1-element collection
    def __getitem__(self, i):
        if i != 0:
            raise IndexError(i)
        return self

    def __len__(self):
        return 1

    def __repr__(self):  # Control.jl:155-157
        return f"{type(self).__name__} with {self.N_coeff} control coefficients and final time tf={self.tf}"


class GRAPEControl(AbstractControl):
    """Piecewise-constant control (grape_control.jl:18-99)."""

    def __init__(self, N_amplitudes: int, tf: float):
        self.N_amplitudes = int(N_amplitudes)
        self.N_coeff = 2 * self.N_amplitudes
        self.tf = float(tf)

    def find_region_index(self, t):
        """0-based region index (grape_control.jl:82-99)."""
        t = np.asarray(t, float)
        if np.any(t < 0) or np.any(t > self.tf * (1 + np.finfo(float).eps)):
            raise ValueError("Value is outside the interval [0,tf]")
        width = self.tf / self.N_amplitudes
        return np.minimum(np.floor(t / width).astype(np.int64), self.N_amplitudes - 1)

    def grad_tables(self, times, dmax):
        times = np.asarray(times, float)
        gp = np.zeros((len(times), dmax + 1, self.N_coeff))
        gq = np.zeros_like(gp)
        idx = self.find_region_index(times)
        rows = np.arange(len(times))
        gp[rows, 0, idx] = 1.0
        gq[rows, 0, idx + self.N_amplitudes] = 1.0
        return gp, gq


def _clamped_uniform_knots(degree: int, n_basis: int) -> np.ndarray:
    """Full knot vector on [0,1] (FortranBSpline.jl:49-57)."""
    k = degree + 1
    n_knots = n_basis + k
    n_distinct = n_knots - 2 * (k - 1)
    if n_distinct < 2:
        raise ValueError("need at least degree+1 basis functions")
    inner = np.arange(n_distinct) / (n_distinct - 1)
    return np.concatenate([np.zeros(k - 1), inner, np.ones(k - 1)])


def bspline_basis_derivatives(degree: int, n_basis: int, x: np.ndarray, dmax: int) -> np.ndarray:
    """All ``n_basis`` clamped uniform B-splines of the given degree on [0,1] and
    their derivatives 0..dmax at the points ``x``: ``[len(x), dmax+1, n_basis]``.

    Cox-de Boor recursion for the values of every lower degree, then the
    derivative recurrence  B'_{i,p} = p (B_{i,p-1}/(t_{i+p}-t_i) - B_{i+1,p-1}/(t_{i+p+1}-t_{i+1})).
    The interval is chosen as the reference does (FortranBSpline.jl:268-277):
    right-continuous, with the last interval closed at x = 1.
    """
    x = np.asarray(x, float)
    p, k = degree, degree + 1
    T = _clamped_uniform_knots(degree, n_basis)
    n_knots = len(T)
    n_distinct = n_knots - 2 * (k - 1)
    span = np.minimum(np.floor(x * (n_distinct - 1) + k).astype(np.int64), n_knots - k) - 1  # 0-based
    nx = len(x)
    # values[q] : degree-q basis functions B_{i,q}, i = 0 .. n_knots-q-2
    values = [np.zeros((nx, n_knots - 1))]
    values[0][np.arange(nx), span] = 1.0
    for q in range(1, p + 1):
        prev = values[q - 1]
        nb = n_knots - q - 1
        cur = np.zeros((nx, nb))
        for i in range(nb):
            d1 = T[i + q] - T[i]
            d2 = T[i + q + 1] - T[i + 1]
            if d1 > 0:
                cur[:, i] += (x - T[i]) / d1 * prev[:, i]
            if d2 > 0:
                cur[:, i] += (T[i + q + 1] - x) / d2 * prev[:, i + 1]
        values.append(cur)

    out = np.zeros((nx, dmax + 1, n_basis))
    out[:, 0, :] = values[p]

    def deriv(q: int, d: int) -> np.ndarray:
        """d-th derivative of the degree-q basis, [nx, n_knots-q-1]."""
        if d == 0:
            return values[q]
        if q == 0:
            return np.zeros((nx, n_knots - 1))
        lower = deriv(q - 1, d - 1)
        nb = n_knots - q - 1
        cur = np.zeros((nx, nb))
        for i in range(nb):
            d1 = T[i + q] - T[i]
            d2 = T[i + q + 1] - T[i + 1]
            if d1 > 0:
                cur[:, i] += q * lower[:, i] / d1
            if d2 > 0:
                cur[:, i] -= q * lower[:, i + 1] / d2
        return cur

    for d in range(1, min(dmax, p) + 1):
        out[:, d, :] = deriv(p, d)
    return out


class FortranBSplineControl(AbstractControl):
    """Clamped uniform B-spline control with ``N_basis_functions`` coefficients for
    p and as many for q (FortranBSpline.jl:17-60; evaluation :86-189).  The
    reference evaluates the basis with pppack's bsplvd through ccall; here it is
    evaluated with :func:`bspline_basis_derivatives`."""

    def __init__(self, degree: int, N_basis_functions: int, tf: float):
        self.degree = int(degree)
        self.N_basis_functions = int(N_basis_functions)
        self.tf = float(tf)
        self.N_coeff = 2 * self.N_basis_functions
        self.bspline_order = self.degree + 1
        self.N_knots = self.N_basis_functions + self.bspline_order
        self.N_distinct_knots = self.N_knots - 2 * (self.bspline_order - 1)
        self.knot_vector = _clamped_uniform_knots(self.degree, self.N_basis_functions)

    def grad_tables(self, times, dmax):
        times = np.asarray(times, float)
        nb = self.N_basis_functions
        B = bspline_basis_derivatives(self.degree, nb, times / self.tf, dmax)
        scale = self.tf ** -np.arange(dmax + 1, dtype=float)  # chain rule, FortranBSpline.jl:98
        B = B * scale[None, :, None]
        gp = np.zeros((len(times), dmax + 1, self.N_coeff))
        gq = np.zeros_like(gp)
        gp[:, :, :nb] = B
        gq[:, :, nb:] = B
        return gp, gq


class GeneralBSplineControl(FortranBSplineControl):
    """Same spline space parameterised by the number of distinct knots
    (GeneralBSplineControl.jl:1-19: N_coeff = 2*(order + N_knots - 2))."""

    def __init__(self, degree: int, N_knots: int, tf: float):
        super().__init__(degree, int(degree) + int(N_knots) - 1, tf)
        self.N_knots_distinct = int(N_knots)


class CarrierControl(AbstractControl):
    """Base control modulated by carrier waves (CarrierControl.jl:5-192):
    p(t) = sum_f cos(w_f t) p_f(t) - sin(w_f t) q_f(t),
    q(t) = sum_f sin(w_f t) p_f(t) + cos(w_f t) q_f(t)."""

    def __init__(self, base_control: AbstractControl, carrier_frequencies: Sequence[float]):
        self.base_control = base_control
        self.carrier_frequencies = np.asarray(carrier_frequencies, float)
        self.N_coeffs_per_frequency = base_control.N_coeff
        self.N_coeff = self.N_coeffs_per_frequency * len(self.carrier_frequencies)
        self.tf = base_control.tf

    def grad_tables(self, times, dmax):
        times = np.asarray(times, float)
        bp, bq = self.base_control.grad_tables(times, dmax)
        npf = self.N_coeffs_per_frequency
        gp = np.zeros((len(times), dmax + 1, self.N_coeff))
        gq = np.zeros_like(gp)
        for f, w in enumerate(self.carrier_frequencies):
            cs, sn = np.cos(w * times), np.sin(w * times)
            sl = slice(f * npf, (f + 1) * npf)
            for d in range(dmax + 1):
                for kk in range(d + 1):
                    wk = w ** kk
                    # kk-th derivatives of cos and -sin (p) / sin and cos (q): CarrierControl.jl:49-61,:78-90
                    dcos = [cs, -sn, -cs, sn][kk % 4] * wk
                    dsin = [sn, cs, -sn, -cs][kk % 4] * wk
                    bc = math.comb(d, kk)
                    gp[:, d, sl] += bc * (dcos[:, None] * bp[:, d - kk, :] - dsin[:, None] * bq[:, d - kk, :])
                    gq[:, d, sl] += bc * (dsin[:, None] * bp[:, d - kk, :] + dcos[:, None] * bq[:, d - kk, :])
        return gp, gq


class BSpline2Control(AbstractControl):
    """Hard-coded quadratic B-spline ("Juqbox" spline, bspline_control.jl:21-249): ``D1`` coefficients
    for p followed by ``D1`` for q, uniform knots with spacing tf/(D1-2), three overlapping pieces per
    point.  Derivatives of order > 2 vanish (bspline2, :139-205)."""

    def __init__(self, D1: int, tf: float):
        D1 = int(D1)
        if D1 < 3:
            raise ValueError(f"Number of coefficients per spline (D1 = {D1}) must be >= 3.")   # :31-33
        self.D1, self.tf = D1, float(tf)
        self.Nseg = 2
        self.N_coeff = 2 * D1
        self.dtknot = self.tf / (D1 - 2)
        self.tcenter = self.dtknot * (np.arange(1, D1 + 1) - 1.5)

    def _basis(self, times, dmax):
        """[len(times), dmax+1, D1]: the three non-zero pieces at every time (bspline2, :151-201)."""
        times = np.asarray(times, float)
        B = np.zeros((len(times), dmax + 1, self.D1))
        width = 3 * self.dtknot
        k = np.clip(np.ceil(times / self.dtknot + 2).astype(np.int64), 3, self.D1)     # 1-based, :149-150
        rows = np.arange(len(times))
        for piece, kk in enumerate((k, k - 1, k - 2)):
            tau = (times - self.tcenter[kk - 1]) / width
            vals = [(9 / 8 + 4.5 * tau + 4.5 * tau ** 2, (4.5 + 9 * tau) / width, np.full_like(tau, 9 / width ** 2)),
                    (0.75 - 9 * tau ** 2, -18 * tau / width, np.full_like(tau, -18 / width ** 2)),
                    (9 / 8 - 4.5 * tau + 4.5 * tau ** 2, (-4.5 + 9 * tau) / width, np.full_like(tau, 9 / width ** 2))][piece]
            for d in range(min(dmax, 2) + 1):
                B[rows, d, kk - 1] = vals[d]
        return B

    def grad_tables(self, times, dmax):
        B = self._basis(times, dmax)
        gp = np.zeros((B.shape[0], dmax + 1, self.N_coeff))
        gq = np.zeros_like(gp)
        gp[:, :, :self.D1] = B
        gq[:, :, self.D1:] = B
        return gp, gq


def BSplineControl(tf: float, D1: int, omega: Sequence[float]):
    """BSplineControl(tf, D1, omega) (bspline_control.jl:257-268): quadratic B-spline envelopes times carrier
    waves in the Juqbox ``bcarrier2`` layout (bspline_backend.jl:783-848): per frequency ``D1``
    coefficients of the cos/p envelope then ``D1`` of the sin/q envelope,
    p = sum_f b1 cos(w t) - b2 sin(w t), q = sum_f b1 sin(w t) + b2 cos(w t) -- which is exactly
    CarrierControl over BSpline2Control."""
    return CarrierControl(BSpline2Control(D1, tf), list(omega))


class HermiteControl(AbstractControl):
    """Piecewise Hermite interpolation (hermite_control.jl:1-283): on each of the ``N_points-1`` equal
    intervals p is the polynomial of degree ``2 N_derivatives + 1`` that matches the value and the first
    ``N_derivatives`` derivatives given at both end points.  The control vector holds, for p and then
    for q, ``[1+N_derivatives, N_points]`` (derivative order fastest) numbers which are turned into
    scaled Taylor coefficients ``dt^j p^(j)/j!`` by ``scaling_type`` (:217-234): "Taylor" 1,
    "Derivative" ``dt^j/j!`` (the parameters are the derivatives), "Heuristic" ``(j+1)! 2^j``."""

    def __init__(self, N_points: int, tf: float, N_derivatives: int, scaling_type: str = "Heuristic"):
        if int(N_points) <= 1:
            raise ValueError("N_points must be > 1")                                   # :44
        self.N_points, self.tf, self.N_derivatives = int(N_points), float(tf), int(N_derivatives)
        self.scaling_type = str(scaling_type).lstrip(":")
        if self.scaling_type not in ("Taylor", "Derivative", "Heuristic"):
            raise ValueError(self.scaling_type)                                        # :229
        m = self.N_derivatives
        self.N_coeff = self.N_points * (m + 1) * 2
        self.dt = self.tf / (self.N_points - 1)
        j = np.arange(m + 1)
        fact = np.array([math.factorial(int(x)) for x in j], dtype=float)
        self.scaling = {"Taylor": np.ones(m + 1), "Derivative": self.dt ** j / fact,
                        "Heuristic": np.array([math.factorial(int(x) + 1) for x in j], dtype=float) * 2.0 ** j}[self.scaling_type]
        # monomial coefficients of the interpolant in z = (t - t_left)/dt from the Taylor data at z = 0, 1
        n = 2 * m + 2
        A = np.zeros((n, n))
        for jj in range(m + 1):
            A[jj, jj] = 1.0
            for k in range(jj, n):
                A[m + 1 + jj, k] = math.comb(k, jj)
        self._Minv = np.linalg.inv(A)

    def _basis(self, times, dmax):
        """[len(times), dmax+1, N_points*(m+1)] for one of p/q."""
        times = np.asarray(times, float)
        m, n = self.N_derivatives, 2 * self.N_derivatives + 2
        if np.any(times < 0) or np.any(times > self.tf * (1 + np.finfo(float).eps)):
            raise ValueError("Value is outside the interval [0,tf]")
        reg = np.minimum(np.floor(times / self.dt).astype(np.int64), self.N_points - 2)   # find_region_index
        z = (times - reg * self.dt) / self.dt
        B = np.zeros((len(times), dmax + 1, self.N_points * (m + 1)))
        rows = np.arange(len(times))
        k = np.arange(n)
        for d in range(min(dmax, n - 1) + 1):
            fall = np.array([math.factorial(int(x)) / math.factorial(int(x) - d) if x >= d else 0.0 for x in k])
            v = fall[None, :] * np.where(k[None, :] >= d, z[:, None] ** np.maximum(k[None, :] - d, 0), 0.0)
            w = v @ self._Minv / self.dt ** d                     # weights on [left data | right data]
            for jj in range(m + 1):
                B[rows, d, reg * (m + 1) + jj] = w[:, jj] * self.scaling[jj]
                B[rows, d, (reg + 1) * (m + 1) + jj] = w[:, m + 1 + jj] * self.scaling[jj]
        return B

    def grad_tables(self, times, dmax):
        B = self._basis(times, dmax)
        half = self.N_coeff // 2
        gp = np.zeros((B.shape[0], dmax + 1, self.N_coeff))
        gq = np.zeros_like(gp)
        gp[:, :, :half] = B
        gq[:, :, half:] = B
        return gp, gq


def HermiteCarrierControl(N_points: int, tf: float, N_derivatives: int, carrier_wave_freqs: Sequence[float],
                          scaling_type: str = "Heuristic"):
    """HermiteCarrierControl (hermite_carrier.jl:1-200): per carrier one block of Hermite-control
    coefficients, p = sum_f h^p_f cos(w t) - h^q_f sin(w t), q = sum_f h^p_f sin(w t) + h^q_f cos(w t)
    -- carrier waves over HermiteControl."""
    return CarrierControl(HermiteControl(N_points, tf, N_derivatives, scaling_type), list(carrier_wave_freqs))


class ZeroControl(AbstractControl):
    """p = q = 0 with ``N_coeff`` inert coefficients (zero_control.jl:1-12)."""

    def __init__(self, N_coeff: int, tf: float):
        self.N_coeff, self.tf = int(N_coeff), float(tf)

    def grad_tables(self, times, dmax):
        z = np.zeros((len(np.asarray(times)), dmax + 1, self.N_coeff))
        return z, z.copy()


class GeneralGRAPEControl(GRAPEControl):
    """Piecewise monomials (generalized_grape_control.jl:1-34): in region r,
    p = pcof[r] * local_t**monomial_order with local_t in [0,1); monomial_order = 0 is GRAPE."""

    def __init__(self, N_amplitudes: int, tf: float, monomial_order: int):
        super().__init__(N_amplitudes, tf)
        self.monomial_order = int(monomial_order)

    def grad_tables(self, times, dmax):
        times = np.asarray(times, float)
        gp = np.zeros((len(times), dmax + 1, self.N_coeff))
        gq = np.zeros_like(gp)
        idx = self.find_region_index(times)
        width = self.tf / self.N_amplitudes
        local = (times - width * idx) / width
        rows = np.arange(len(times))
        mo = self.monomial_order
        for d in range(min(dmax, mo) + 1):
            v = math.factorial(mo) / math.factorial(mo - d) * local ** (mo - d) / width ** d
            gp[rows, d, idx] = v
            gq[rows, d, idx + self.N_amplitudes] = v
        return gp, gq


class _TrigControl(AbstractControl):
    """Two-coefficient trigonometric controls (sincos_control.jl:1-97): p = f_p(w t) pcof[0],
    q = f_q(w t) pcof[1] with f in {sin, cos}."""
    _p_phase = 0.0   # sin(x + phase): 0 -> sin, pi/2 -> cos
    _q_phase = 0.0
    N_coeff = 2

    def __init__(self, tf: float, frequency: float = 1.0):
        self.tf, self.frequency = float(tf), float(frequency)

    def grad_tables(self, times, dmax):
        times = np.asarray(times, float)
        gp = np.zeros((len(times), dmax + 1, self.N_coeff))
        gq = np.zeros_like(gp)
        w = self.frequency
        for d in range(dmax + 1):
            gp[:, d, 0] = w ** d * np.sin(w * times + self._p_phase + d * np.pi / 2)
            if self.N_coeff > 1:
                gq[:, d, 1] = w ** d * np.sin(w * times + self._q_phase + d * np.pi / 2)
        return gp, gq


class SinCosControl(_TrigControl):
    _p_phase, _q_phase = 0.0, np.pi / 2


class SinControl(_TrigControl):
    _p_phase, _q_phase = 0.0, 0.0


class CosControl(_TrigControl):
    _p_phase, _q_phase = np.pi / 2, np.pi / 2


class SingleSymCosControl(_TrigControl):
    """p = cos(w t) pcof[0], q = 0 (sincos_control.jl:99-115)."""
    _p_phase = np.pi / 2
    N_coeff = 1


# ---------------------------------------------------------------------------
# collections of controls (Control.jl:67-96)
# ---------------------------------------------------------------------------
def as_control_list(controls):
    if isinstance(controls, AbstractControl):
        return [controls]
    return list(controls)


def get_number_of_control_parameters(controls) -> int:
    return sum(c.N_coeff for c in as_control_list(controls))


def get_control_vector_slice(pcof, controls, control_index: int):
    """View of pcof belonging to control ``control_index`` (0-based here)."""
    cl = as_control_list(controls)
    start = sum(c.N_coeff for c in cl[:control_index])
    return pcof[start:start + cl[control_index].N_coeff]


def control_tables_general(controls, pcof, nsteps: int, tf: float, n_deriv: int):
    """Any ``AbstractControl`` (Control.jl:6-27), linear in its coefficients or not, through its pointwise protocol:
    the tables ``p[d, k, n] = p_k^(d)(t_n)/d!`` (what ``fill_p_mat!`` gives, Control.jl:125-149, stacked over the time
    grid) and their Jacobians ``Gp[k][n, d, l] = d/dpcof_l`` of the same entries AT THIS pcof
    (``eval_grad_p_derivative!``), which is what the discrete adjoint contracts its per-time-point scalars with.
    Returns ``(p, q, Gp, Gq)``; ``p, q`` are ``[n_deriv+1, n_controls, nsteps+1]`` (Fortran order)."""
    cl = as_control_list(controls)
    pcof = np.asarray(pcof, float)
    dt = tf / nsteps
    nt = nsteps + 1
    p = np.zeros((n_deriv + 1, len(cl), nt), order="F")
    q = np.zeros_like(p, order="F")
    Gp, Gq, off = [], [], 0
    for k, c in enumerate(cl):
        th = pcof[off:off + c.N_coeff]
        gp = np.zeros((nt, n_deriv + 1, c.N_coeff)); gq = np.zeros_like(gp)
        for n in range(nt):
            t = n * dt
            for d in range(n_deriv + 1):
                f = 1.0 / math.factorial(d)
                p[d, k, n] = f * c.eval_p_derivative(t, th, d)
                q[d, k, n] = f * c.eval_q_derivative(t, th, d)
                gp[n, d] = f * np.asarray(c.eval_grad_p_derivative(t, th, d), float)
                gq[n, d] = f * np.asarray(c.eval_grad_q_derivative(t, th, d), float)
        Gp.append(gp); Gq.append(gq)
        off += c.N_coeff
    return p, q, Gp, Gq


def fill_p_mat(vals_mat, controls, t, pcof):
    """vals_mat[d, k] = p_k^(d)(t)/d!  (Control.jl:125-136)."""
    for k, c in enumerate(as_control_list(controls)):
        c.fill_p_vec(vals_mat[:, k], t, get_control_vector_slice(pcof, controls, k))
    return vals_mat


def fill_q_mat(vals_mat, controls, t, pcof):
    for k, c in enumerate(as_control_list(controls)):
        c.fill_q_vec(vals_mat[:, k], t, get_control_vector_slice(pcof, controls, k))
    return vals_mat


def control_basis(controls, nsteps: int, tf: float, n_deriv: int):
    """Basis tensors for a whole time grid.

    Returns ``(Gp, Gq, offsets)``: ``Gp[k]`` is ``[nsteps+1, n_deriv+1, N_coeff_k]``
    with ``Gp[k][n, d, l] = d/dpcof_l ( p_k^(d)(t_n) / d! )``, time grid
    ``t_n = n*(tf/nsteps)`` exactly as the stepper forms it
    (forward_evolution.jl:166,190).  ``offsets[k]`` is where control k's slice
    starts in pcof."""
    cl = as_control_list(controls)
    dt = tf / nsteps
    times = np.arange(nsteps + 1) * dt
    inv_fact = np.array([1.0 / math.factorial(d) for d in range(n_deriv + 1)])
    Gp, Gq, offsets, off = [], [], [], 0
    for c in cl:
        gp, gq = c.grad_tables(times, n_deriv)
        Gp.append(np.ascontiguousarray(gp * inv_fact[None, :, None]))
        Gq.append(np.ascontiguousarray(gq * inv_fact[None, :, None]))
        offsets.append(off)
        off += c.N_coeff
    return Gp, Gq, offsets
