"""BASELINE.json configs[4] at full size on ONE GPU: synthetic 4-qudit random SchrodingerProb
(N=256, 256 columns, 4 control operators, order 12, tf=2, nsteps=200).  Prints the per-phase device
times and a directional finite-difference check of the gradient (size-independent property)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa: F401
from __graft_entry__ import import_package
import cases

qgd = import_package()
prob, ctrl, pcof, target = cases.synthetic_case(qgd, N=256, c=256, n_ops=4, nsteps=200, tf=2.0)
target = (prob.u0 + 1j * prob.v0)            # any fixed target; N_ess = N
t0 = time.time()
dp = qgd.DeviceProblem(prob, 12)
dp.set_controls(ctrl); dp.set_target(target); dp.set_timing(1)
print(f"setup {time.time() - t0:.1f} s")
for it in range(2):
    t0 = time.time()
    grad, out3 = dp.discrete_adjoint(pcof)
    print(f"evaluation {it}: {time.time() - t0:.3f} s  -> {200 / (time.time() - t0):.0f} timesteps/s")
print({k: round(v, 2) for k, v in sorted(dp.timings().items(), key=lambda kv: -kv[1])})
d = np.random.default_rng(1).standard_normal(len(pcof)); d /= np.linalg.norm(d)
eps = 1e-3      # the objective is ~ -3e4 here (random, un-normalised U0 as target): rounding noise 1e-11 / eps; scripts/c5_determinism.py
def obj(p):
    a, b, g = dp.eval_forward(p)
    return 1 - (a * a + b * b) / prob.N_ess_levels ** 2 + g
fd = (obj(pcof + eps * d) - obj(pcof - eps * d)) / (2 * eps)
print("directional derivative: adjoint %.10e  central difference %.10e  rel diff %.2e" % (grad @ d, fd, abs(grad @ d - fd) / abs(fd)))
