import os, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
from __graft_entry__ import import_package
import cases
qgd = import_package()
prob, ctrl, pcof, target = cases.synthetic_case(qgd, N=144, c=72, n_ops=2, nsteps=40, tf=0.4, seed=3)
rng = np.random.default_rng(3)
pcs = [pcof * (1.0 + 0.3 * rng.standard_normal(len(pcof))) for _ in range(6)]
os.environ["QGD_GRAPH"] = "1"
dp = qgd.DeviceProblem(prob, 8); dp.set_controls(ctrl); dp.set_target(target)
got = [dp.discrete_adjoint(p) for p in pcs]
dp.close()
del os.environ["QGD_GRAPH"]
worst = 0
for i in (0, 3, 5):
    fresh = qgd.DeviceProblem(prob, 8); fresh.set_controls(ctrl); fresh.set_target(target)
    g, o = fresh.discrete_adjoint(pcs[i]); fresh.close()
    worst = max(worst, np.abs(g - got[i][0]).max() / np.abs(g).max(), np.abs(np.asarray(o) - np.asarray(got[i][1])).max())
print("graph vs plain worst", worst)
