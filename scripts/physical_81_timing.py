"""Four dispersively coupled qutrits (N = 81, 16 columns, guard levels, order 8, 400 steps): the reference's kind of problem
one size beyond N = 64, where the operators are sparse but the N > 64 path treats them as dense."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from __graft_entry__ import import_package
qgd = import_package()
nsteps, order = 400, 8
sizes, ess = (3, 3, 3, 3), (2, 2, 2, 2)
freqs = 2 * np.pi * np.array([4.10, 4.35, 4.60, 4.85])
kerr = 2 * np.pi * np.array([[0.20, 0.004, 0.003, 0.002], [0.004, 0.22, 0.005, 0.003], [0.003, 0.005, 0.21, 0.004], [0.002, 0.003, 0.004, 0.19]])
prob = qgd.DispersiveProblem(sizes, ess, freqs, freqs, kerr, 0.5 * nsteps, nsteps)
ctrl = [qgd.CarrierControl(qgd.FortranBSplineControl(2, 12, prob.tf), [0.0, -float(kerr[k, k])]) for k in range(prob.N_operators)]
rng = np.random.default_rng(81)
pcof = 0.05 * (rng.random(qgd.get_number_of_control_parameters(ctrl)) - 0.5)
N, c = prob.N_tot_levels, prob.N_initial_conditions
target = np.linalg.qr(rng.standard_normal((N, c)) + 1j * rng.standard_normal((N, c)))[0]
dp = qgd.DeviceProblem(prob, order); dp.set_controls(ctrl); dp.set_target(target)
dp.set_timing(1)
for _ in range(3): dp.discrete_adjoint(pcof)
tm = dp.timings()
dp.set_timing(0)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): dp.discrete_adjoint(pcof)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
print(f"N={N} c={c} order {order} nsteps {nsteps}: {dt*1e3:.3f} ms per evaluation, {nsteps/dt:.0f} timesteps/s  path {dp.operator_path()}")
print({k: round(v, 3) for k, v in sorted(tm.items(), key=lambda kv: -kv[1])})
