"""cnot3 (N = 64, 8 columns, 550 steps) against the Hermite order 2 .. 16: looks for performance cliffs between the
template instantiations of the sparse kernels.  python scripts/order_sweep_timing.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from __graft_entry__ import import_package
import cases
qgd = import_package()
for order in (2, 4, 6, 8, 10, 12, 14, 16):
    prob, ctrl, pcof, target = cases.cnot3_case(qgd, nsteps=550, tf=550.0)
    dp = qgd.DeviceProblem(prob, order); dp.set_controls(ctrl); dp.set_target(target)
    dp.set_timing(1)
    for _ in range(3): dp.discrete_adjoint(pcof)
    tm = dp.timings()
    dp.set_timing(0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): dp.discrete_adjoint(pcof)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    top = sorted(tm.items(), key=lambda kv: -kv[1])[:5]
    print(f"order {order:2d}: {dt*1e3:7.3f} ms   " + "  ".join(f"{k} {v:.3f}" for k, v in top), flush=True)
    dp.close(); qgd.clear_cache()
