"""build_LR phase time of the cnot3 problem against the number of time points: 1024 workgroups (nt = 512) are exactly two rounds of
the 512 resident slots of k_build_LR_ell, nt = 513 starts a third (development check; HIP events around the phase)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from __graft_entry__ import import_package
import bench
qgd = import_package()
for nsteps in (255, 256, 383, 384, 511, 512, 550, 767, 768):
    prob, ctrl, pcof, target = bench.workload(qgd, nsteps, float(nsteps))
    dp = qgd.DeviceProblem(prob, 8); dp.set_controls(ctrl); dp.set_target(target)
    dp.set_timing(1)
    acc = {}
    for i in range(14):
        dp.discrete_adjoint(pcof)
        if i >= 4:
            for k, v in dp.timings().items(): acc[k] = acc.get(k, 0) + v / 10
    print(nsteps + 1, "time points:", {k: round(v * 1e3, 1) for k, v in acc.items() if k in ("build_LR", "inverse", "gradient", "lambda", "tables")}, flush=True)
    dp.close()
