"""Tiny time grids (1, 2, 3 steps) through every kernel family against the numpy statement."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from __graft_entry__ import import_package
import cases, proto_propagator as pp
qgd = import_package()
worst = 0.0
for N, c in ((4, 4), (20, 3), (64, 8), (80, 8), (144, 144), (272, 16)):
    for nsteps in (1, 2, 3):
        for order in (2, 8):
            prob, ctrl, pcof, target = cases.synthetic_case(qgd, N=N, c=c, n_ops=2, nsteps=nsteps, tf=0.01 * nsteps, seed=N + nsteps)
            Gp, Gq, off = qgd.control_basis(ctrl, prob.nsteps, prob.tf, order // 2)
            ref = pp.evaluate(prob, Gp, Gq, off, pcof, target, order)
            dp = qgd.DeviceProblem(prob, order); dp.set_controls(ctrl); dp.set_target(target)
            hist = np.zeros(dp._hist_shape(), order="F")
            g, o = dp.discrete_adjoint(pcof, False, hist)
            dp.close(); qgd.clear_cache()
            href = pp.history_real(ref["ws"])
            e = max(np.abs(hist - href).max() / np.abs(href).max(), np.abs(g - ref["grad"]).max() / max(np.abs(ref["grad"]).max(), 1e-300))
            worst = max(worst, e)
            print(f"N={N} c={c} nsteps={nsteps} order={order}: {e:.1e}", flush=True)
print("worst", worst)
assert worst < 1e-10
