"""Phase times of the single-kernel path (library built with -DQGD_TINY_PROFILE: QGD_LIB_PATH=scripts/ubench/bin/libqgd_prof.so)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from __graft_entry__ import import_package
import cases
qgd = import_package()
order = int(sys.argv[1]) if len(sys.argv) > 1 else 4
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 90
prob, ctrl, pcof, target = cases.cnot2_case(qgd, nsteps=nsteps, tf=float(nsteps))
dp = qgd.DeviceProblem(prob, order); dp.set_controls(ctrl); dp.set_target(target); dp.set_timing(0)
for _ in range(3): dp.discrete_adjoint(pcof)
torch.cuda.synchronize()
print("small path taken:", dp.small_path_taken())
