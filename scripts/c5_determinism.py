"""Config 5 forward evaluation: the overlaps <w_N,R>, <w_N,T> and the directional derivative printed to full
precision, to compare algebraically equivalent kernel paths (environment switches) at rounding level."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from __graft_entry__ import import_package
import cases
qgd = import_package()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
prob, ctrl, pcof, target = cases.synthetic_case(qgd, N=N, c=N, n_ops=4, nsteps=200, tf=2.0)
target = prob.u0 + 1j * prob.v0
dp = qgd.DeviceProblem(prob, 12); dp.set_controls(ctrl); dp.set_target(target)
a, b, g = dp.eval_forward(pcof)
grad, _ = dp.discrete_adjoint(pcof)
d = np.random.default_rng(1).standard_normal(len(pcof)); d /= np.linalg.norm(d)
print(f"a={a!r} b={b!r} guard={g!r} grad.d={grad @ d!r}")
for eps in (1e-3, 1e-4, 1e-5):
    f = []
    for sgn in (1, -1):
        aa, bb, gg = dp.eval_forward(pcof + sgn * eps * d)
        f.append(1 - (aa * aa + bb * bb) / prob.N_ess_levels ** 2 + gg)
    print(f"eps={eps:g}: fd={(f[0] - f[1]) / (2 * eps)!r}")
