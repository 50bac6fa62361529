"""Actual (not just within-tolerance) error of the large-N device path against the numpy statement of the
algorithm: final state, stage derivatives and gradient for N (argv[1], default 100), 32 columns, orders 4 and 12."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa: F401
from __graft_entry__ import import_package
import cases, proto_propagator as pp

qgd = import_package()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
runs = ((4, 40), (12, 40)) if N <= 128 else ((4, 12), (12, 8))
for order, nsteps in runs:
    prob, ctrl, pcof, target = cases.synthetic_case(qgd, N=N, c=32, nsteps=nsteps, tf=0.01 * nsteps)
    Gp, Gq, off = qgd.control_basis(ctrl, prob.nsteps, prob.tf, order // 2)
    ref = pp.evaluate(prob, Gp, Gq, off, pcof, target, order)
    dp = qgd.DeviceProblem(prob, order)
    dp.set_controls(ctrl); dp.set_target(target)
    grad, _ = dp.discrete_adjoint(pcof)
    hist = np.zeros((2 * N, order // 2 + 1, nsteps + 1, 32), order="F")
    qgd.eval_forward_(hist, prob, ctrl, pcof, order=order)
    href = pp.history_real(ref["ws"])
    for j in range(order // 2 + 1):
        e = np.abs(hist[:, j] - href[:, j]).max() / np.abs(href[:, j]).max()
        print(f"order {order}: stage derivative {j}: max rel err {e:.2e}")
    e_n = [np.abs(hist[:, 0, n] - href[:, 0, n]).max() for n in (1, 2, 5, nsteps)]
    print(f"N={N} order {order}: state error at steps 1,2,5,{nsteps}:", " ".join(f"{x:.1e}" for x in e_n))
    print(f"order {order}: gradient max rel err {np.abs(grad - ref['grad']).max() / np.abs(ref['grad']).max():.2e}")
