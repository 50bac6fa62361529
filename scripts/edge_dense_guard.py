"""A dense (non-diagonal) guard projector on the N > 64 path against the numpy statement."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from __graft_entry__ import import_package
import cases, proto_propagator as pp
qgd = import_package()
worst = 0.0
for N, c, diag in ((80, 8, False), (80, 8, True), (144, 32, False), (32, 4, False)):
    prob, ctrl, pcof, target = cases.synthetic_case(qgd, N=N, c=c, n_ops=2, nsteps=12, tf=0.12, seed=N)
    rng = np.random.default_rng(N)
    if diag:
        W = np.diag(rng.random(2 * N))
    else:
        B = rng.standard_normal((2 * N, 2 * N)) / np.sqrt(2 * N); W = B @ B.T
    prob.guard_subspace_projector = np.asfortranarray(W)
    order = 8
    Gp, Gq, off = qgd.control_basis(ctrl, prob.nsteps, prob.tf, order // 2)
    ref = pp.evaluate(prob, Gp, Gq, off, pcof, target, order)
    dp = qgd.DeviceProblem(prob, order); dp.set_controls(ctrl); dp.set_target(target)
    g, o = dp.discrete_adjoint(pcof)
    dp.close(); qgd.clear_cache()
    e = max(np.abs(g - ref["grad"]).max() / np.abs(ref["grad"]).max(), abs(o[2] - ref["guard"]) / max(1.0, abs(ref["guard"])))
    worst = max(worst, e)
    print(f"N={N} c={c} diag={diag}: gradient/guard err {e:.1e}  guard {o[2]:.6e}")
assert worst < 1e-10
