"""cnot3 evaluation against the length of the time grid (dt = 1 throughout): timesteps/s and the per-phase device
times.  The headline grid (550 steps) leaves every kernel latency-bound -- 2.15 inverse workgroups per CU, scan blocks
on a quarter of the CUs; longer grids show what the same kernels reach when the batch fills the chip."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from __graft_entry__ import import_package
import bench
qgd = import_package()
for nsteps in (138, 275, 550, 1100, 2200, 4400, 8800):
    prob, ctrl, pcof, target = bench.workload(qgd, nsteps, float(nsteps))
    dp = qgd.DeviceProblem(prob, 8); dp.set_controls(ctrl); dp.set_target(target)
    dp.set_timing(1)
    for _ in range(3): dp.discrete_adjoint(pcof)
    ph = dp.timings()
    dp.set_timing(0)
    for _ in range(30): dp.discrete_adjoint(pcof)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    reps = max(10, 11000 // nsteps)
    for _ in range(reps): dp.discrete_adjoint(pcof)
    torch.cuda.synchronize(); el = (time.perf_counter() - t0) / reps
    inv = ph.get("inverse", 0.0)
    tf = (nsteps * 2 * 8 * 64 ** 3) / (inv * 1e-3) / 1e12 if inv else 0.0
    keys = ("tables", "build_LR", "inverse", "sweep_forward", "sweep_forward2", "sweep_adjoint", "sweep_adjoint2", "lambda", "gradient")
    print(f"nsteps {nsteps:5d}: {el * 1e3:7.3f} ms  {nsteps / el / 1e6:5.2f} M timesteps/s   inverse {inv * 1e3:6.1f} us = {tf:4.1f} TFLOP/s ({tf / 78.6:.2f})   "
          + " ".join(f"{k}={ph.get(k, 0) * 1e3:.0f}" for k in keys), flush=True)
    dp.close(); qgd.clear_cache()
