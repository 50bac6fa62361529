"""Time per cnot3 evaluation over a long back-to-back run, in chunks of 20 (does the rate depend on how long the GPU has
been busy? -- clock ramp-up after idle, power management under sustained fp64 MFMA load)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from __graft_entry__ import import_package
import bench
qgd = import_package()
prob, ctrl, pcof, target = bench.workload(qgd, 550, 550.0)
dp = qgd.DeviceProblem(prob, 8); dp.set_controls(ctrl); dp.set_target(target); dp.set_timing(0)
dp.discrete_adjoint(pcof); torch.cuda.synchronize()
time.sleep(float(sys.argv[1]) if len(sys.argv) > 1 else 2.0)        # let the GPU go idle first
if len(sys.argv) > 2 and sys.argv[2] == "nogc":
    import gc; gc.collect(); gc.disable()
out = []
for chunk in range(40):
    t0 = time.perf_counter()
    for _ in range(20): dp.discrete_adjoint(pcof)
    out.append((time.perf_counter() - t0) / 20 * 1e6)
print(" ".join(f"{v:.0f}" for v in out))
