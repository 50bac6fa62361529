"""Full-size cnot3 (550 steps, 180 parameters): forced gradient vs discrete adjoint on the device, timed."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import torch, numpy as np
from __graft_entry__ import import_package
qgd = import_package()
import cases
prob, target = qgd.cnot3_problem(nsteps=550, tf=550.0)
ctrl = cases.cnot3_controls(qgd, prob)
pcof = (np.random.default_rng(0).random(qgd.get_number_of_control_parameters(ctrl)) - 0.5) * 2 * np.pi * 0.005
dp = qgd.DeviceProblem(prob, 8); dp.set_controls(ctrl); dp.set_target(target); dp.set_timing(1)
g_adj, _ = dp.discrete_adjoint(pcof)
g_for = dp.eval_grad_forced(pcof)
t0 = time.time(); K = 5
for _ in range(K): g_for = dp.eval_grad_forced(pcof)
t_for = (time.time() - t0) / K
t0 = time.time()
for _ in range(20): dp.discrete_adjoint(pcof)
t_adj = (time.time() - t0) / 20
print(f"forced vs adjoint: max rel diff {np.abs(g_for - g_adj).max() / np.abs(g_adj).max():.2e}; forced {t_for * 1e3:.2f} ms, adjoint {t_adj * 1e3:.3f} ms per evaluation")
print({k: round(v, 3) for k, v in dp.timings().items()})
