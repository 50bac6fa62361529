"""Race detector of last resort: the same cnot3 gradient evaluation many times; every result must agree with the
first to rounding (the only run-to-run freedom is the order of the atomic adds of the scalars)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from __graft_entry__ import import_package
import cases
qgd = import_package()
worst = 0.0
for which, kw, order in (("cnot3", dict(nsteps=550, tf=550.0), 8), ("cnot3", dict(nsteps=97, tf=48.5), 4), ("guarded", dict(nsteps=90, tf=45.0), 6), ("cnot2", dict(nsteps=100, tf=100.0), 8)):
    prob, ctrl, pcof, target = getattr(cases, which + "_case")(qgd, **kw)
    dp = qgd.DeviceProblem(prob, order); dp.set_controls(ctrl); dp.set_target(target)
    g0, o0 = dp.discrete_adjoint(pcof)
    dev = 0.0
    for i in range(400):
        g, o = dp.discrete_adjoint(pcof)
        dev = max(dev, np.abs(g - g0).max() / np.abs(g0).max(), np.abs(o - o0).max() / max(1.0, np.abs(o0).max()))
    print(f"{which} {kw} order {order}: max relative deviation over 400 evaluations {dev:.2e}")
    worst = max(worst, dev)
    dp.close()
assert worst < 1e-12
print("ok")
