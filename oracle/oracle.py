"""ctypes binding of the CPU ORACLE (oracle/libqgd_oracle.so).

TEST INFRASTRUCTURE ONLY.  May be imported by tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg -- never by the product package.  See
qgd_oracle.h for what the oracle restates and how it is pinned.

The functions take the host-side mirror objects (SchrodingerProb, controls) by
duck typing and return numpy arrays in the reference's (Julia, column-major)
layouts.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libqgd_oracle.so")


def build(force: bool = False) -> str:
    """Compile the oracle (and oracle/_ref when /root/reference is present)."""
    src = os.path.join(_HERE, "qgd_oracle.c")
    stale = (not os.path.exists(_LIB_PATH)) or os.path.getmtime(_LIB_PATH) < max(
        os.path.getmtime(src), os.path.getmtime(os.path.join(_HERE, "qgd_oracle.h")))
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "oracle"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference/src/Fortran") and not os.path.exists(os.path.join(_HERE, "_ref", "bspline_lib.so")):
        subprocess.call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return _LIB_PATH


class _Control(C.Structure):
    pass


_Control._fields_ = [
    ("kind", C.c_int32), ("n_coeff", C.c_int32), ("tf", C.c_double),
    ("n_amplitudes", C.c_int32), ("degree", C.c_int32), ("n_basis", C.c_int32),
    ("n_freq", C.c_int32), ("freqs", C.POINTER(C.c_double)), ("base", C.POINTER(_Control)),
]


class _Prob(C.Structure):
    _fields_ = [
        ("N", C.c_int32), ("n_ops", C.c_int32), ("n_cols", C.c_int32), ("n_ess", C.c_int32),
        ("nsteps", C.c_int32), ("precond", C.c_int32), ("tf", C.c_double),
        ("gmres_abstol", C.c_double), ("gmres_reltol", C.c_double),
        ("system_sym", C.c_void_p), ("system_asym", C.c_void_p),
        ("sym_ops", C.c_void_p), ("asym_ops", C.c_void_p),
        ("u0", C.c_void_p), ("v0", C.c_void_p), ("guard", C.c_void_p), ("csc", C.c_void_p),
    ]


class _Csc(C.Structure):
    _fields_ = [("colptr", C.c_void_p), ("rowval", C.c_void_p), ("nzval", C.c_void_p)]


_SPARSE = False


def set_sparse_operators(on: bool):
    """Operators as SparseMatrixCSC (the reference's DispersiveProblem default, sparse_rep=true) instead of dense."""
    global _SPARSE
    _SPARSE = bool(on)


class _Stats(C.Structure):
    _fields_ = [("fwd_gmres_iters", C.c_double), ("adj_gmres_iters", C.c_double), ("applies", C.c_int64)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        alt = os.environ.get("QGD_ORACLE_LIB")        # another build of the same source (scripts/oracle_sanitizers.sh: ASan + UBSan)
        if not alt:
            build()
        _lib = C.CDLL(alt or _LIB_PATH)
        _lib.qo_coefficient.restype = C.c_double
        _lib.qo_coefficient.argtypes = [C.c_int, C.c_int, C.c_int]
        _lib.qo_eval_p_derivative.restype = C.c_double
        _lib.qo_eval_q_derivative.restype = C.c_double
        _lib.qo_infidelity_real.restype = C.c_double
        _lib.qo_guard_penalty_real.restype = C.c_double
        _lib.qo_bspline_basis_derivs.restype = C.c_int
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _dptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class Problem:
    """Keeps the numpy buffers alive next to the C struct."""

    def __init__(self, prob, precond=None):
        f = lambda a: np.asfortranarray(np.array(a, dtype=np.float64))
        self.bufs = dict(
            system_sym=f(prob.system_sym), system_asym=f(prob.system_asym),
            sym_ops=np.ascontiguousarray(np.stack([f(o).T for o in prob.sym_operators])) if prob.N_operators else np.zeros(1),
            asym_ops=np.ascontiguousarray(np.stack([f(o).T for o in prob.asym_operators])) if prob.N_operators else np.zeros(1),
            u0=f(prob.u0), v0=f(prob.v0), guard=f(prob.guard_subspace_projector))
        # sym_ops[k] stored as C-contiguous transpose == column-major original
        if precond is None:
            precond = 1 if getattr(prob, "preconditioner_type", "") == "DiagonalHamiltonianPreconditioner" else 0
        csc_ptr = None
        if _SPARSE:
            from scipy.sparse import csc_matrix
            mats = [prob.system_sym, prob.system_asym, *prob.sym_operators, *prob.asym_operators]
            self.csc_arrays = []
            self.csc = (_Csc * len(mats))()
            for i, a in enumerate(mats):
                sp = csc_matrix(np.asarray(a))
                arrs = (sp.indptr.astype(np.int64), sp.indices.astype(np.int64), sp.data.astype(np.float64))
                self.csc_arrays.append(arrs)
                self.csc[i] = _Csc(_ptr(arrs[0]), _ptr(arrs[1]), _ptr(arrs[2]))
            csc_ptr = C.cast(self.csc, C.c_void_p)
        self.c = _Prob(prob.N_tot_levels, prob.N_operators, prob.N_initial_conditions, prob.N_ess_levels,
                       prob.nsteps, precond, prob.tf, prob.gmres_abstol, prob.gmres_reltol,
                       *[_ptr(self.bufs[k]) for k in ("system_sym", "system_asym", "sym_ops", "asym_ops", "u0", "v0", "guard")],
                       csc_ptr)
        self.N = prob.N_tot_levels
        self.n_cols = prob.N_initial_conditions
        self.nsteps = prob.nsteps


class Controls:
    def __init__(self, controls):
        from_list = [controls] if not isinstance(controls, (list, tuple)) else list(controls)
        self.keep = []
        self.structs = [self._convert(c) for c in from_list]
        self.arr = (C.POINTER(_Control) * len(self.structs))(*[C.pointer(s) for s in self.structs])
        self.n_pcof = sum(c.N_coeff for c in from_list)

    def _convert(self, c):
        name = type(c).__name__
        s = _Control()
        s.n_coeff = c.N_coeff
        s.tf = c.tf
        if name == "GRAPEControl":
            s.kind = 0
            s.n_amplitudes = c.N_amplitudes
        elif name in ("FortranBSplineControl", "GeneralBSplineControl"):
            s.kind = 1
            s.degree = c.degree
            s.n_basis = c.N_basis_functions
        elif name == "BSpline2Control":       # hard-coded quadratic B-spline (bspline_control.jl:21-249)
            s.kind = 3
            s.n_basis = c.D1
        elif name == "BCarrier2Control":      # the reference's BSplineControl in its own bcarrier2 formulas (orders 0, 1)
            s.kind = 4
            fr = np.ascontiguousarray(c.omega, dtype=np.float64)
            self.keep.append(fr)
            s.n_basis = c.D1
            s.n_freq = len(fr)
            s.freqs = _dptr(fr)
        elif name == "CarrierControl":
            s.kind = 2
            fr = np.ascontiguousarray(c.carrier_frequencies, dtype=np.float64)
            base = self._convert(c.base_control)
            self.keep += [fr, base]
            s.n_freq = len(fr)
            s.freqs = _dptr(fr)
            s.base = C.pointer(base)
        else:
            raise TypeError(f"oracle has no restatement of control type {name}")
        self.keep.append(s)
        return s


class BCarrier2Control:
    """BSplineControl(tf, D1, omega) of the reference (bspline_control.jl:251-395) for the oracle: evaluated by the
    restated bcarrier2 / bcarrier2_dt / gradbcarrier2 / gradbcarrier2_dt formulas (bspline_backend.jl:381-955), NOT through
    the carrier-wave wrapper -- the independent statement the product's CarrierControl(BSpline2Control) is checked against.
    Derivative orders 0 and 1 only, as in the reference (Hermite order 2)."""

    def __init__(self, tf, D1, omega):
        self.tf, self.D1, self.omega = float(tf), int(D1), [float(w) for w in omega]
        self.N_coeff = 2 * self.D1 * len(self.omega)


def set_num_threads(n: int):
    lib().qo_set_num_threads(int(n))


def set_converged_terminal(on: bool):
    """See qgd_oracle.c: default False = the reference's gmres! defaults for the terminal solve."""
    lib().qo_set_converged_terminal(1 if on else 0)


def set_parallel_gradient(on: bool):
    """Large test problems: the per-column gradient accumulation on threads, summed in column order -- bit-identical to
    the reference's serial loop (qgd_oracle.c: qo_set_parallel_gradient).  Never on for the cpu_baseline."""
    lib().qo_set_parallel_gradient(1 if on else 0)


COST_TYPES = {"Infidelity": 0, ":Infidelity": 0, "Tracking": 1, ":Tracking": 1, "Norm": 2, ":Norm": 2}


def set_cost_type(cost_type="Infidelity"):
    """cost_type of discrete_adjoint / eval_grad_forced / eval_grad_finite_difference (a process-wide switch like the
    one above; eval_grad_discrete_adjoint.jl:26-35)."""
    lib().qo_set_cost_type(COST_TYPES[cost_type])


def coefficient(j, p, q):
    return lib().qo_coefficient(j, p, q)


def target_real(target):
    t = np.asarray(target)
    return np.asfortranarray(np.vstack([np.real(t), np.imag(t)]).astype(np.float64))


def fill_p_vec(control, t, pcof, nvals, q=False):
    cs = Controls(control)
    out = np.zeros(nvals)
    pc = np.ascontiguousarray(pcof, dtype=np.float64)
    fn = lib().qo_fill_q_vec if q else lib().qo_fill_p_vec
    fn(C.byref(cs.structs[0]), C.c_double(t), _dptr(pc), C.c_int(nvals), _dptr(out))
    return out


def eval_grad_derivative(control, t, pcof, order, q=False):
    cs = Controls(control)
    out = np.zeros(control.N_coeff)
    pc = np.ascontiguousarray(pcof, dtype=np.float64)
    fn = lib().qo_eval_grad_q_derivative if q else lib().qo_eval_grad_p_derivative
    fn(C.byref(cs.structs[0]), C.c_double(t), _dptr(pc), C.c_int(order), _dptr(out))
    return out


def bspline_basis_derivs(degree, n_basis, x, nderiv):
    k = degree + 1
    out = np.zeros((nderiv, k))
    first = lib().qo_bspline_basis_derivs(C.c_int(degree), C.c_int(n_basis), C.c_double(x), C.c_int(nderiv), _dptr(out))
    return first, out  # out[d, i]


def recursive_magic(prob, controls, pcof, m, control_index, t, w_mat, lam, deriv_order, coeff):
    """One recursive_magic! call: coeff * <d w_k/d theta_l, lambda> for the coefficients l of one control."""
    P = Problem(prob)
    cs = Controls(controls)
    pc = np.ascontiguousarray(pcof, dtype=np.float64)
    w = np.asfortranarray(np.array(w_mat, dtype=np.float64))
    la = np.ascontiguousarray(lam, dtype=np.float64)
    out = np.zeros(cs.structs[control_index].n_coeff)
    lib().qo_recursive_magic(C.byref(P.c), cs.arr, _dptr(pc), C.c_int(m), C.c_int(control_index), C.c_double(t), _dptr(w),
                             _dptr(la), C.c_int(deriv_order), C.c_double(coeff), _dptr(out))
    return out


def compute_derivatives(prob, pvals, qvals, uv, forcing=None, adjoint=False):
    """uv: [2N, 1+m] with column 0 set; returns filled copy.  pvals/qvals: [(1+m), n_ops]."""
    P = Problem(prob)
    uv = np.asfortranarray(np.array(uv, dtype=np.float64))
    m = uv.shape[1] - 1
    pv = np.asfortranarray(np.array(pvals, dtype=np.float64))
    qv = np.asfortranarray(np.array(qvals, dtype=np.float64))
    if adjoint:
        lib().qo_compute_adjoint_derivatives(C.byref(P.c), _dptr(pv), _dptr(qv), C.c_int(m), _dptr(uv))
    else:
        fo = None if forcing is None else np.asfortranarray(np.array(forcing, dtype=np.float64))
        lib().qo_compute_derivatives(C.byref(P.c), _dptr(pv), _dptr(qv), C.c_int(m),
                                     None if fo is None else _dptr(fo), _dptr(uv))
    return uv


def eval_forward(prob, controls, pcof, order=2, forcing=None, return_stats=False, precond=None):
    P, Cs = Problem(prob, precond), Controls(controls)
    m = order // 2
    hist = np.zeros((2 * P.N, 1 + m, 1 + P.nsteps, P.n_cols), order="F")
    pc = np.ascontiguousarray(pcof, dtype=np.float64)
    fo = None if forcing is None else np.asfortranarray(np.array(forcing, dtype=np.float64))
    st = _Stats()
    lib().qo_eval_forward(C.byref(P.c), Cs.arr, _dptr(pc), C.c_int(order),
                          None if fo is None else _dptr(fo), _dptr(hist), C.byref(st))
    return (hist, st.fwd_gmres_iters) if return_stats else hist


def eval_adjoint(prob, controls, pcof, terminal_condition, order=2, forcing=None):
    P, Cs = Problem(prob), Controls(controls)
    m = order // 2
    lam = np.zeros((2 * P.N, 1 + m, 1 + P.nsteps, P.n_cols), order="F")
    pc = np.ascontiguousarray(pcof, dtype=np.float64)
    term = np.asfortranarray(np.array(terminal_condition, dtype=np.float64).reshape(2 * P.N, P.n_cols, order="F"))
    fo = None if forcing is None else np.asfortranarray(np.array(forcing, dtype=np.float64))
    lib().qo_eval_adjoint(C.byref(P.c), Cs.arr, _dptr(pc), C.c_int(order), _dptr(term),
                          None if fo is None else _dptr(fo), _dptr(lam), None)
    return lam


def infidelity_real(psi, target_r, n_ess):
    psi = np.asfortranarray(np.array(psi, dtype=np.float64))
    tr = np.asfortranarray(np.array(target_r, dtype=np.float64))
    if psi.ndim == 1:
        psi, tr = psi.reshape(-1, 1), tr.reshape(-1, 1)
    return lib().qo_infidelity_real(C.c_int(psi.shape[0] // 2), C.c_int(psi.shape[1]), _dptr(psi), _dptr(tr), C.c_int(n_ess))


def guard_penalty_real(prob, history):
    P = Problem(prob)
    h = np.asfortranarray(history)
    return lib().qo_guard_penalty_real(C.byref(P.c), C.c_int(h.shape[1] - 1), _dptr(h))


def discrete_adjoint(prob, controls, pcof, target, order=2, history=None, return_all=False, precond=None):
    """Returns grad (and history, lambda_history, adjoint_forcing, stats when return_all)."""
    P, Cs = Problem(prob, precond), Controls(controls)
    m = order // 2
    shape = (2 * P.N, 1 + m, 1 + P.nsteps, P.n_cols)
    pre = history is not None
    hist = np.asfortranarray(history) if pre else np.zeros(shape, order="F")
    lam = np.zeros(shape, order="F")
    forcing = np.zeros((2 * P.N, 1 + P.nsteps, P.n_cols), order="F")
    pc = np.ascontiguousarray(pcof, dtype=np.float64)
    tr = target_real(target)
    grad = np.zeros(Cs.n_pcof)
    st = _Stats()
    lib().qo_discrete_adjoint(C.byref(P.c), Cs.arr, _dptr(pc), C.c_int(Cs.n_pcof), _dptr(tr), C.c_int(order),
                              C.c_int(1 if pre else 0), _dptr(grad), _dptr(hist), _dptr(lam), _dptr(forcing), C.byref(st))
    if return_all:
        return grad, hist, lam, forcing, st
    return grad


def eval_grad_forced(prob, controls, pcof, target, order=2):
    P, Cs = Problem(prob), Controls(controls)
    pc = np.ascontiguousarray(pcof, dtype=np.float64)
    tr = target_real(target)
    grad = np.zeros(Cs.n_pcof)
    lib().qo_eval_grad_forced(C.byref(P.c), Cs.arr, _dptr(pc), C.c_int(Cs.n_pcof), _dptr(tr), C.c_int(order), _dptr(grad))
    return grad


def eval_grad_finite_difference(prob, controls, pcof, target, order=2, dpcof=1e-5):
    P, Cs = Problem(prob), Controls(controls)
    pc = np.ascontiguousarray(pcof, dtype=np.float64)
    tr = target_real(target)
    grad = np.zeros(Cs.n_pcof)
    lib().qo_eval_grad_finite_difference(C.byref(P.c), Cs.arr, _dptr(pc), C.c_int(Cs.n_pcof), _dptr(tr),
                                         C.c_int(order), C.c_double(dpcof), _dptr(grad))
    return grad


def apply_hamiltonian(prob, pvals, qvals, w, derivative_order=0, use_adjoint=False):
    """out = (+/-) A_d w for every column of w (2N x c); pvals/qvals: [(1+m), n_ops] tables."""
    P = Problem(prob)
    pv = np.asfortranarray(np.array(pvals, dtype=np.float64))
    qv = np.asfortranarray(np.array(qvals, dtype=np.float64))
    w = np.asfortranarray(np.array(w, dtype=np.float64))
    out = np.zeros_like(w, order="F")
    for c in range(w.shape[1]):
        wi = np.ascontiguousarray(w[:, c]); oi = np.zeros(w.shape[0])
        lib().qo_apply_hamiltonian(C.byref(P.c), _dptr(pv), _dptr(qv), C.c_int(pv.shape[0]),
                                   C.c_int(derivative_order), C.c_int(1 if use_adjoint else 0), _dptr(wi), _dptr(oi))
        out[:, c] = oi
    return out
