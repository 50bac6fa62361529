"""1500 create / evaluate / destroy cycles of handles (cnot2 on the small-problem path, every fourth one cnot3 at 550 steps on the
fused front) in one process: looks for leaks and lifecycle bugs.  gpurun -- python scripts/handle_lifecycle_stress.py"""
import os, sys, time, faulthandler
faulthandler.enable()
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as ge
qgd = ge.import_package()
import cases
p2 = cases.cnot2_case(qgd)
p3 = cases.cnot3_case(qgd, nsteps=550, tf=550.0)
t0 = time.time()
for i in range(1500):
    prob, ctrl, pcof, target = p2 if i % 4 else p3
    dp = qgd.DeviceProblem(prob, 8); dp.set_controls(ctrl); dp.set_target(target)
    g, o = dp.discrete_adjoint(pcof); f = dp.eval_forward(pcof)
    dp.close()
    if i % 100 == 0: print(i, round(time.time() - t0, 1), flush=True)
print("done")
