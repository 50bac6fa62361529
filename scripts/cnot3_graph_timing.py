"""cnot3 headline problem, event bracketing off: wall time per evaluation with the hipGraph replay
(QGD_GRAPH=1) and with plain launches (default)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from __graft_entry__ import import_package
import cases
qgd = import_package()
prob, ctrl, pcof, target = cases.cnot3_case(qgd, nsteps=550, tf=550.0)
dp = qgd.DeviceProblem(prob, 8); dp.set_controls(ctrl); dp.set_target(target)
for _ in range(6): dp.discrete_adjoint(pcof)
ts = []
for _ in range(200):
    t1 = time.perf_counter(); dp.discrete_adjoint(pcof); ts.append(time.perf_counter() - t1)
ts = np.array(ts) * 1e6
print(f"graph {'on' if os.environ.get('QGD_GRAPH') else 'off'}: median {np.median(ts):.1f} us per evaluation -> {550 / np.median(ts) * 1e6:.0f} timesteps/s (min {ts.min():.1f})")
