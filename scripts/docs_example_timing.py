"""The reference's documented two-qudit example (docs/src/examples.md:91-131: DispersiveProblem (4,4)/(2,2), N = 16, 4 initial
conditions, tf = 50, 100 steps, two BSplineControl's with two carriers each): microseconds per gradient evaluation at orders
2-8, and per phase at order 4 -- the size between the four-launch small-problem path (N <= 4) and the sizes where the launches
are filled.  The general path runs it as twelve dependent launches of one 16 x 16 MFMA tile each."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from __graft_entry__ import import_package
qgd = import_package()
tf, nsteps = 50.0, 100
freqs = np.array([4.10595, 4.81526]) * 2 * np.pi
kerr = 2 * np.pi * np.array([[2 * 0.1099, 0.1], [0.1, 2 * 0.1126]])
prob = qgd.DispersiveProblem((4, 4), (2, 2), freqs, freqs.copy(), kerr, tf, nsteps, sparse_rep=True)
om = [2 * np.pi * np.array([0.0, -kerr[0, 1]]), 2 * np.pi * np.array([0.0, -kerr[0, 1]])]
ctrl = [qgd.BSplineControl(tf, 10, om[0]), qgd.BSplineControl(tf, 10, om[1])]
npar = qgd.get_number_of_control_parameters(ctrl)
pcof = np.random.default_rng(0).random(npar) * 0.004
target = qgd.create_gate((4, 4), (2, 2), [((1, 0), (1, 1)), ((1, 1), (1, 0))])
print(f"N = {prob.N_tot_levels}, columns {prob.N_initial_conditions}, P = {npar}")
for order in (2, 4, 8):
    dp = qgd.DeviceProblem(prob, order); dp.set_controls(ctrl); dp.set_target(target)
    for _ in range(20): g, o = dp.discrete_adjoint(pcof)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): g, o = dp.discrete_adjoint(pcof)
    torch.cuda.synchronize(); el = (time.perf_counter() - t0) / 200
    print(f"order {order}: {el * 1e6:.1f} us per gradient evaluation ({nsteps / el / 1e6:.2f} M timesteps/s), infidelity {1 - (o[0] ** 2 + o[1] ** 2) / 16:.4f}, path {dp.operator_path()}")
    if order == 4:
        dp.set_timing(1)
        for _ in range(5): dp.discrete_adjoint(pcof)
        print("   phases (us, device events):", {k: round(v * 1e3, 1) for k, v in sorted(dp.timings().items(), key=lambda kv: -kv[1])})
        dp.set_timing(0)
    dp.close()
