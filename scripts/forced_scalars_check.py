"""out3 of qgd_eval_forward_forced with a ZERO forcing must equal out3 of qgd_eval_forward (guard penalty included)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from __graft_entry__ import import_package
import cases
qgd = import_package()
for which, order, kw in (("guarded", 6, {}), ("cnot3", 8, dict(nsteps=40, tf=20.0)), ("dense_guard", 6, {})):
    prob, ctrl, pcof, target = getattr(cases, which + "_case")(qgd, **kw)
    dp = qgd.DeviceProblem(prob, order); dp.set_controls(ctrl); dp.set_target(target)
    a = np.asarray(dp.eval_forward(pcof))
    z = np.zeros((prob.real_system_size, order // 2, prob.nsteps + 1, prob.N_initial_conditions), order="F")
    b = np.asarray(dp.eval_forward_forced(pcof, z))
    print(which, "plain", a, "forced(zero)", b, "diff", np.abs(a - b).max())
    dp.close()
