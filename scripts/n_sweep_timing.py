"""One evaluation (dense random operators, 2 controls, order 8, 100 steps) against N: looks for performance cliffs between
the kernel families (N <= 64 MFMA panels, N > 64 GEMM tiles, generic fallbacks).  python scripts/n_sweep_timing.py [c]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from __graft_entry__ import import_package
import cases
qgd = import_package()
c = int(sys.argv[1]) if len(sys.argv) > 1 else 8
order = int(sys.argv[2]) if len(sys.argv) > 2 else 8
for N in (2, 4, 8, 9, 16, 17, 24, 32, 33, 48, 49, 63, 64, 65, 72, 80, 96, 100, 112, 128, 129, 160, 192, 200, 240, 256, 257, 272, 288, 289, 304, 320, 400, 512, 592):
    cc = min(c, N)
    prob, ctrl, pcof, target = cases.synthetic_case(qgd, N=N, c=cc, n_ops=2, nsteps=100, tf=1.0)
    dp = qgd.DeviceProblem(prob, order); dp.set_controls(ctrl); dp.set_target(target)
    dp.set_timing(1)
    for _ in range(2): dp.discrete_adjoint(pcof)
    tm = dp.timings()
    dp.set_timing(0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(4): dp.discrete_adjoint(pcof)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 4
    top = sorted(tm.items(), key=lambda kv: -kv[1])[:3]
    print(f"N={N:4d} c={cc:3d}: {dt*1e3:8.3f} ms   " + "  ".join(f"{k} {v:.3f}" for k, v in top), flush=True)
    dp.close(); qgd.clear_cache()
