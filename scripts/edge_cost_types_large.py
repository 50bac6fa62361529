"""cost_type :Tracking / :Norm / :Infidelity on a large panel (N = 144, 144 columns: the many-workgroup terminal kernels with
their fixed-order overlap sums) against the single-workgroup terminal kernel (QGD_TERMINAL_ONE_WG=1)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from __graft_entry__ import import_package
import cases
qgd = import_package()
prob, ctrl, pcof, target = cases.synthetic_case(qgd, N=144, c=144, n_ops=2, nsteps=10, tf=0.1, seed=4)
worst = 0.0
for cost in ("Infidelity", "Tracking", "Norm"):
    res = {}
    for one in (False, True):
        if one: os.environ["QGD_TERMINAL_ONE_WG"] = "1"
        else: os.environ.pop("QGD_TERMINAL_ONE_WG", None)
        dp = qgd.DeviceProblem(prob, 8); dp.set_controls(ctrl); dp.set_target(target); dp.set_cost_type(cost)
        g, o = dp.discrete_adjoint(pcof)
        g2, o2 = dp.discrete_adjoint(pcof)
        assert np.array_equal(g, g2) and np.array_equal(np.asarray(o)[:2], np.asarray(o2)[:2])      # bitwise from run to run
        dp.close()
        res[one] = (g, np.asarray(o))
    e = max(np.abs(res[0][0] - res[1][0]).max() / np.abs(res[1][0]).max(), np.abs(res[0][1] - res[1][1]).max() / max(1.0, np.abs(res[1][1]).max()))
    worst = max(worst, e)
    print(cost, "many-workgroup vs one-workgroup terminal:", f"{e:.1e}", " scalars", res[0][1])
os.environ.pop("QGD_TERMINAL_ONE_WG", None)
assert worst < 1e-12
