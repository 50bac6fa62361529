"""What the live sampling of the dominant kernel costs bench.py's timed region (cnot3 headline evaluation):
plain loop / event pair on every 8th step / the same plus reading the timings inside the loop."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from __graft_entry__ import import_package
import bench
qgd = import_package()
prob, ctrl, pcof, target = bench.workload(qgd, 550, 550.0)
dp = qgd.DeviceProblem(prob, 8); dp.set_controls(ctrl); dp.set_target(target); dp.set_timing(0)
for _ in range(5): dp.discrete_adjoint(pcof)
K = 40
def run(mode):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(K):
        s = (i % 8 == 0) and mode > 0
        if mode > 0: dp.set_timing(2 if s else 0, "inverse")
        dp.discrete_adjoint(pcof)
        if s and mode > 1: dp.timings()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e6
for rep in range(2):
    for mode, name in ((0, "plain"), (1, "event pair every 8th"), (2, "+ timings() in the loop"), (3, "every step + timings")):
        if mode == 3:
            torch.cuda.synchronize(); t0 = time.perf_counter()
            dp.set_timing(2, "inverse")
            for i in range(K):
                dp.discrete_adjoint(pcof); dp.timings()
            torch.cuda.synchronize(); el = (time.perf_counter() - t0) / K * 1e6
            dp.set_timing(0)
        else:
            el = run(mode)
        print(f"{name:32s} {el:7.1f} us")
