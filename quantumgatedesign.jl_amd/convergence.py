"""Step-doubling convergence studies, mirroring
src/Tests/test_convergence.jl:20-146 (``get_histories``) and :233-250 (Richardson extrapolation):
the report generator the reference uses for its accuracy-versus-time plots, driven here by the
device ``eval_forward``.  Results are plain dictionaries of numpy arrays; ``save_histories`` /
``load_histories`` store them as JLD2/HDF5 (a ``.jld2`` name: ``jld2io``, through libhdf5) or as ``.npz``.
"""
from __future__ import annotations

import time
from collections import OrderedDict

import numpy as np

from . import jld2io
from .evolution import eval_forward, release


def richardson_extrap_sol(A_h, A_2h, order):
    """Order ``order+1`` solution from solutions with step h and 2h (test_convergence.jl:244-250)."""
    n = order
    return ((2 ** n) * A_h - A_2h) / (2 ** n - 1)


def richardson_extrap_rel_err(A_h, A_2h, order):
    """Estimate of the relative error of ``A_h`` (test_convergence.jl:233-241), Frobenius norms."""
    sol = richardson_extrap_sol(A_h, A_2h, order)
    return float(np.linalg.norm(sol - A_h) / np.linalg.norm(sol))


def get_histories(prob, controls, pcof, N_iterations, orders=(2, 4, 6, 8, 10), min_error_limit=-np.inf,
                  max_error_limit=-np.inf, base_nsteps=None, nsteps_change_factor=2, start_iteration=1,
                  filename=None, quiet=False):
    """Run ``eval_forward`` with ``base_nsteps * nsteps_change_factor**(k-1)`` steps, k = start_iteration..
    N_iterations, for every order; every run stores the time points of the coarsest grid
    (``saveEveryNsteps`` = the step multiplier), so consecutive histories can be compared point by
    point.  Returns ``{"Order p (QGD)": {order, nsteps, step_sizes, elapsed_times, histories,
    richardson_errors}}`` with the reference's early-exit rules (precision reached / numerical
    saturation)."""
    say = (lambda *a: None) if quiet else print
    base = prob.nsteps if base_nsteps is None else int(base_nsteps)
    work = prob.copy()
    ret = OrderedDict()
    try:
        return _run_histories(work, controls, pcof, N_iterations, orders, min_error_limit, max_error_limit, base,
                              nsteps_change_factor, start_iteration, filename, say, ret)
    finally:
        release(work)      # one device grid per order, sized for the finest time grid: give them back now


def _run_histories(work, controls, pcof, N_iterations, orders, min_error_limit, max_error_limit, base,
                   nsteps_change_factor, start_iteration, filename, say, ret):
    for order in orders:
        summary = dict(order=order, nsteps=[], step_sizes=[], elapsed_times=[], histories=[], richardson_errors=[])
        ret[f"Order {order} (QGD)"] = summary
        for k in range(start_iteration, N_iterations + 1):
            mult = nsteps_change_factor ** (k - 1)
            work.nsteps = base * mult
            t0 = time.time()
            history = eval_forward(work, controls, pcof, order=order, saveEveryNsteps=mult)
            elapsed = time.time() - t0
            err = float("nan")
            if summary["histories"]:
                err = richardson_extrap_rel_err(history, summary["histories"][-1], order)
            summary["nsteps"].append(work.nsteps)
            summary["step_sizes"].append(work.tf / work.nsteps)
            summary["elapsed_times"].append(elapsed)
            summary["histories"].append(history)
            summary["richardson_errors"].append(err)
            say(f"order {order:2d}  nsteps {work.nsteps:8d}  Richardson error {err:.3e}  elapsed {elapsed:.3f} s")
            if filename is not None:
                save_histories(ret, filename)
            errs = summary["richardson_errors"]
            if errs[-1] < min_error_limit:
                say("Breaking early due to precision reached")
                break
            if len(errs) > 2 and errs[-1] < max_error_limit and errs[-1] > errs[-2] > errs[-3]:
                say("Breaking early due to numerical saturation")
                break
    return ret


def observed_orders(summary):
    """log2 of the ratio of consecutive Richardson error estimates: the observed convergence order
    (what test/ForwardEvolutionTests/forward_convergence.jl:47-65 asserts against the nominal one)."""
    e = np.asarray(summary["richardson_errors"], dtype=float)
    step = np.asarray(summary["step_sizes"], dtype=float)
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.log(e[1:-1] / e[2:]) / np.log(step[1:-1] / step[2:])


def save_histories(ret, filename):
    """The dictionary of get_histories on disk.  ``.jld2``: one group per "Order k (QGD)" summary with the reference's
    keys (src/Tests/test_convergence.jl:60-81; ``histories`` as a Vector of 4-d arrays), written through libhdf5
    (``jld2io``) -- JLD2 itself stores each summary as a serialised Dict object, which only Julia can produce; the keys
    and array encodings inside are the same.  Other names: a flat ``.npz``."""
    if jld2io.is_jld2_name(filename):
        jld2io.save(filename, {name: {k: (list(v) if k == "histories" else np.asarray(v)) for k, v in summary.items()}
                               for name, summary in ret.items()})
        return
    flat = {}
    for name, summary in ret.items():
        for key, val in summary.items():
            if key == "histories":
                for i, h in enumerate(val):
                    flat[f"{name}/histories/{i}"] = h
            else:
                flat[f"{name}/{key}"] = np.asarray(val)
    np.savez(filename, **flat)


def load_histories(filename):
    if jld2io.is_jld2_name(filename):
        ret = OrderedDict()
        for name, summary in jld2io.load(filename).items():
            ret[name] = {k: (list(v) if k == "histories" else (v.tolist() if isinstance(v, np.ndarray) else v)) for k, v in summary.items()}
        return ret
    data = np.load(filename if str(filename).endswith(".npz") else str(filename) + ".npz")
    ret = OrderedDict()
    for full in data.files:
        name, key = full.split("/", 1)
        summary = ret.setdefault(name, dict(histories={}))
        if key.startswith("histories/"):
            summary["histories"][int(key.split("/")[1])] = data[full]
        else:
            summary[key] = data[full].tolist() if data[full].ndim else data[full].item()
    for summary in ret.values():
        summary["histories"] = [summary["histories"][i] for i in sorted(summary["histories"])]
    return ret
