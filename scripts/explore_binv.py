import os, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
from __graft_entry__ import import_package
import cases
qgd = import_package()
N = 128
for dt in (4.0, 8.0, 16.0, 32.0, 64.0):
    for order in (4, 8):
        prob, ctrl, pcof, target = cases.synthetic_case(qgd, N=N, c=16, n_ops=2, nsteps=8, tf=8 * dt, seed=77)
        dp = qgd.DeviceProblem(prob, order); dp.set_controls(ctrl); dp.set_target(target)
        try:
            dp.discrete_adjoint(pcof)
        except Exception as e:
            print(dt, order, "error", e); dp.close(); continue
        Linv, L, redone = dp.intermediate("Linv"), dp.intermediate("L"), int(dp.intermediate("repivoted"))
        dp.close()
        res = max(np.abs(Linv[n][:N,:N] @ L[n][:N,:N] - np.eye(N)).max() for n in range(1, 9))
        cond = max(np.linalg.cond(L[n][:N,:N]) for n in range(1, 9))
        c11 = max(np.linalg.cond(L[n][:64,:64]) for n in range(1, 9))
        print(f"dt {dt} order {order}: redone {redone} residual {res:.2e} cond {cond:.2e} cond(L11) {c11:.2e} |L| {np.abs(L[1]).max():.2e}")
