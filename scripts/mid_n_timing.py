"""Mid-size dense problems (64 < N <= 128): per-phase device time of one gradient evaluation.
(Round 2 used it to compare the older LDS-panel kernels, which N > 64 no longer takes, with the GEMM-style kernels of qgd_k_dense.hip.)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from __graft_entry__ import import_package
import cases
qgd = import_package()
for N, c, order, nsteps in ((81, 16, 8, 400), (100, 32, 8, 400), (125, 8, 8, 400), (125, 27, 12, 200)):
    prob, ctrl, pcof, target = cases.synthetic_case(qgd, N=N, c=c, nsteps=nsteps, tf=0.01 * nsteps)
    dp = qgd.DeviceProblem(prob, order); dp.set_controls(ctrl); dp.set_target(target)
    dp.set_timing(1)
    for _ in range(3): g, _ = dp.discrete_adjoint(pcof)
    tm = dp.timings()
    dp.set_timing(0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): dp.discrete_adjoint(pcof)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f"N={N} c={c} order={order} nsteps={nsteps}: {dt*1e3:.2f} ms  |grad|={np.linalg.norm(g):.6e} ", {k: round(v, 2) for k, v in sorted(tm.items(), key=lambda kv: -kv[1])[:7]})
    dp.close()
