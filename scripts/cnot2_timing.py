import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from __graft_entry__ import import_package
import cases
qgd = import_package()
for order in ((8,) if os.environ.get('QGD_CNOT2_ORDER8') else (2, 4, 6, 8, 10)):
    prob, ctrl, pcof, target = cases.cnot2_case(qgd, nsteps=100, tf=100.0)
    dp = qgd.DeviceProblem(prob, order); dp.set_controls(ctrl); dp.set_target(target)
    dp.set_timing(1)
    for _ in range(3): dp.discrete_adjoint(pcof)
    tm = dp.timings()
    dp.set_timing(0)
    for _ in range(5): dp.discrete_adjoint(pcof)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    K = 200
    for _ in range(K): dp.discrete_adjoint(pcof)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
    print(f"cnot2 order {order}: {dt*1e6:.1f} us per evaluation, {prob.nsteps/dt:.0f} timesteps/s; device phases sum {sum(tm.values())*1e3:.1f} us", {k: round(v*1e3,1) for k,v in tm.items()})
    dp.close()
