"""``optimize_gate`` -- the immediate caller of the hot path,
mirroring src/ipopt_optimal_control.jl:187-471.

The reference drives Ipopt (limited-memory BFGS, memory 40, tol 1e-5, box bounds, stop when
the objective drops below 1e-7).  Ipopt is not in this image; the same problem
(box-constrained smooth minimisation with a limited-memory quasi-Newton Hessian) is handed to
scipy's L-BFGS-B with the same memory, bounds and stopping rule.  The objective is
infidelity + guard penalty + ridge penalty (:256-283); every evaluation is one
forward+adjoint pass on the device (``history_precomputed`` reuse is implicit: L-BFGS-B always
asks for value and gradient together).
"""
from __future__ import annotations

import time
from dataclasses import dataclass, field

import numpy as np

from . import jld2io
from .controls import get_number_of_control_parameters
from .evolution import device_problem


@dataclass
class OptimizationHistory:
    """Same fields as the reference's struct (src/ipopt_optimal_control.jl:21-45)."""
    iter_count: list = field(default_factory=list)
    ipopt_obj_value: list = field(default_factory=list)
    wall_time: list = field(default_factory=list)
    pcof: list = field(default_factory=list)
    grad_pcof: list = field(default_factory=list)
    analytic_obj_value: list = field(default_factory=list)
    infidelity: list = field(default_factory=list)
    guard_penalty: list = field(default_factory=list)
    ridge_penalty: list = field(default_factory=list)

    def __len__(self):
        return len(self.iter_count)

    FIELDS = ("iter_count", "ipopt_obj_value", "wall_time", "pcof", "grad_pcof", "analytic_obj_value",
              "infidelity", "guard_penalty", "ridge_penalty")

    def as_dict(self):
        """The nine keys with the Julia types of the reference's struct: Vector{Int64}, Vector{Float64} and, for
        ``pcof`` / ``grad_pcof``, Vector{Vector{Float64}} (a list of arrays)."""
        out = {}
        for k in self.FIELDS:
            v = getattr(self, k)
            if k in ("pcof", "grad_pcof"):
                out[k] = [np.asarray(x, dtype=np.float64) for x in v]
            else:
                out[k] = np.asarray(v, dtype=np.int64 if k == "iter_count" else np.float64)
        return out

    def write(self, filename, setup=None):
        """write(obj::OptimizationHistory, filename) (src/ipopt_optimal_control.jl:74-86): the same nine top-level
        keys.  A name ending in ``.jld2`` gives a JLD2/HDF5 file (``jld2io``: written through libhdf5, the layout
        ``read_optimization_history`` of the reference loads with ``JLD2.load``), with ``setup`` as the ``Setup`` group
        of :223-241; any other name a numpy ``.npz`` archive."""
        if jld2io.is_jld2_name(filename):
            data = self.as_dict()
            if setup:
                data = dict(Setup=setup, **data)
            jld2io.save(filename, data)
            return
        np.savez(filename, **{k: np.asarray(getattr(self, k)) for k in self.FIELDS})

    def __repr__(self):
        if not len(self):
            return "OptimizationHistory\n0 iterations performed."
        i = int(np.argmin(self.infidelity))
        return (f"OptimizationHistory\n{len(self)} iterations performed.\n{self.wall_time[-1]} seconds elapsed.\n"
                f"Minimum infidelity was {self.infidelity[i]}, at iteration {i + 1}.")


def read_optimization_history(filename):
    """read_optimization_history (src/ipopt_optimal_control.jl:91-104): from a JLD2/HDF5 file (a ``.jld2`` name; also
    one written by the reference -- its nine keys are plain arrays and reference arrays) or the ``.npz`` written above."""
    if jld2io.is_jld2_name(filename):
        data = jld2io.load(filename)
        return OptimizationHistory(**{k: [np.asarray(x) for x in data[k]] if k in ("pcof", "grad_pcof") else list(np.asarray(data[k]))
                                      for k in OptimizationHistory.FIELDS})
    data = np.load(filename if str(filename).endswith(".npz") else str(filename) + ".npz")
    return OptimizationHistory(**{k: list(data[k]) for k in OptimizationHistory.FIELDS})


class _Stop(Exception):
    pass


def optimize_gate(schro_prob, controls, pcof_init, target, order=4, pcof_L=None, pcof_U=None, maxIter=50,
                  print_level=5, ridge_penalty_strength=1e-2, max_cpu_time=60.0 * 60 * 24, filename=None):
    """optimize_gate(prob, controls, pcof_init, target; order=4, pcof_L, pcof_U, maxIter=50,
    print_level=5, ridge_penalty_strength=1e-2, max_cpu_time) -> OptimizationHistory."""
    from scipy.optimize import minimize

    pcof_init = np.asarray(pcof_init, dtype=np.float64)
    N_coeff = get_number_of_control_parameters(controls)
    if len(pcof_init) != N_coeff:
        raise ValueError("length of pcof_init does not match the controls")          # :203
    dp = device_problem(schro_prob, order)
    dp.set_controls(controls)
    dp.set_target(target)
    hist = OptimizationHistory()
    last = {}
    t0 = time.time()
    # the one-time entries of the result file (update_jld2, :223-241); the problem and the controls are Julia structs
    # there -- here the plain fields that define them
    setup = dict(target=np.asarray(target, dtype=np.complex128), ridge_penalty_strength=float(ridge_penalty_strength),
                 max_cpu_time=float(max_cpu_time), pcof_init=pcof_init, order=int(order),
                 schrodinger_prob=dict(tf=float(schro_prob.tf), nsteps=int(schro_prob.nsteps),
                                       N_ess_levels=int(schro_prob.N_ess_levels),
                                       N_tot_levels=int(schro_prob.N_tot_levels), N_operators=int(schro_prob.N_operators),
                                       u0=np.asarray(schro_prob.u0), v0=np.asarray(schro_prob.v0)),
                 controls=dict(N_coeff=np.asarray([c.N_coeff for c in (controls if isinstance(controls, (list, tuple)) else [controls])]),
                               kinds=", ".join(type(c).__name__ for c in (controls if isinstance(controls, (list, tuple)) else [controls]))))
    if pcof_L is not None:
        setup["pcof_L"] = np.broadcast_to(np.asarray(pcof_L, float), (N_coeff,)).copy()
    if pcof_U is not None:
        setup["pcof_U"] = np.broadcast_to(np.asarray(pcof_U, float), (N_coeff,)).copy()

    def fun(pcof):
        grad, out3 = dp.discrete_adjoint(pcof)                                       # :257-268, :304
        infid = 1.0 - (out3[0] ** 2 + out3[1] ** 2) / schro_prob.N_ess_levels ** 2   # infidelity.jl:17
        ridge = float(pcof @ pcof) * ridge_penalty_strength / len(pcof)              # :272
        grad = grad + 2.0 * ridge_penalty_strength * pcof / len(pcof)                # :311
        obj = infid + out3[2] + ridge
        last.update(pcof=pcof.copy(), grad=grad.copy(), obj=obj, infid=infid, guard=out3[2], ridge=ridge)
        if time.time() - t0 > max_cpu_time:
            raise _Stop
        return obj, grad

    def callback(xk):                                                                # :348-383
        hist.iter_count.append(len(hist.iter_count))
        hist.ipopt_obj_value.append(last["obj"])
        hist.wall_time.append(time.time() - t0)
        hist.pcof.append(last["pcof"]); hist.grad_pcof.append(last["grad"])
        hist.analytic_obj_value.append(last["obj"]); hist.infidelity.append(last["infid"])
        hist.guard_penalty.append(last["guard"]); hist.ridge_penalty.append(last["ridge"])
        if print_level >= 5:
            print(f"iter {len(hist):4d}  objective {last['obj']:.6e}  infidelity {last['infid']:.6e}  guard {last['guard']:.3e}")
        if filename is not None:                                                    # update_jld2, :223-241
            hist.write(filename, setup)
        if last["obj"] < 1e-7:
            raise _Stop

    def bound(b, default):
        if b is None:
            return [default] * N_coeff
        return list(np.broadcast_to(np.asarray(b, float), (N_coeff,)))

    bounds = list(zip(bound(pcof_L, -np.inf), bound(pcof_U, np.inf)))                # :385-403
    res = None
    try:
        res = minimize(fun, pcof_init, jac=True, method="L-BFGS-B", bounds=bounds, callback=callback,
                       options=dict(maxiter=maxIter, maxcor=40, ftol=1e-15, gtol=1e-9, maxls=30))
    except _Stop:
        pass
    if res is not None and not np.array_equal(res.x, last["pcof"]):
        fun(np.asarray(res.x, dtype=np.float64))       # make `last` describe the returned iterate
    if not len(hist) or not np.array_equal(hist.pcof[-1], last["pcof"]):
        hist.iter_count.append(len(hist.iter_count)); hist.ipopt_obj_value.append(last["obj"])
        hist.wall_time.append(time.time() - t0); hist.pcof.append(last["pcof"])
        hist.grad_pcof.append(last["grad"]); hist.analytic_obj_value.append(last["obj"])
        hist.infidelity.append(last["infid"]); hist.guard_penalty.append(last["guard"])
        hist.ridge_penalty.append(last["ridge"])
    if filename is not None:
        hist.write(filename, setup)
    return hist
