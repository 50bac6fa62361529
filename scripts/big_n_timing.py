import os, sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
from __graft_entry__ import import_package
import cases
qgd = import_package()
for N, c, order, nsteps in ((256, 32, 8, 100), (288, 32, 8, 100), (320, 32, 8, 100), (512, 32, 8, 100)):
    prob, ctrl, pcof, target = cases.synthetic_case(qgd, N=N, c=c, n_ops=2, nsteps=nsteps, tf=0.01 * nsteps)
    dp = qgd.DeviceProblem(prob, order); dp.set_controls(ctrl); dp.set_target(target)
    dp.set_timing(1)
    for _ in range(2): g, _ = dp.discrete_adjoint(pcof)
    tm = dp.timings()
    dp.set_timing(0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): dp.discrete_adjoint(pcof)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    print(f"N={N} c={c} order={order} nsteps={nsteps}: {dt*1e3:.2f} ms ", {k: round(v, 2) for k, v in sorted(tm.items(), key=lambda kv: -kv[1])[:8]}, flush=True)
    dp.close()
