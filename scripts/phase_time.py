"""Per-phase device times (HIP events, every phase bracketed) of the cnot3 headline evaluation, mean of 10 evaluations;
QGD_LIB_PATH selects another build of the library (knock-out / profiling builds)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from __graft_entry__ import import_package
import bench
qgd = import_package()
prob, ctrl, pcof, target = bench.workload(qgd, 550, 550.0)
dp = qgd.DeviceProblem(prob, 8); dp.set_controls(ctrl); dp.set_target(target)
dp.set_timing(1)
acc = {}
for i in range(12):
    try: dp.discrete_adjoint(pcof)
    except Exception as e: pass
    if i >= 2:
        for k, v in dp.timings().items(): acc[k] = acc.get(k, 0) + v / 10
print(os.environ.get("QGD_LIB_PATH", "default"), {k: round(v * 1e3, 1) for k, v in acc.items()})
