"""Shape sweep of the N > 64 kernels against the numpy statement of the algorithm: odd sizes, partial tiles,
1..40 columns, orders 2..16, 0..5 control operators."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from __graft_entry__ import import_package
import cases, proto_propagator as pp
qgd = import_package()
shapes = [(65, 1, 2, 2, 9), (72, 9, 4, 1, 13), (96, 40, 6, 3, 11), (130, 17, 16, 2, 5), (200, 3, 2, 5, 7), (81, 8, 10, 4, 25),
          (113, 24, 8, 3, 30), (128, 33, 14, 1, 6), (160, 16, 12, 2, 8), (255, 5, 4, 3, 6), (290, 12, 6, 2, 5), (100, 7, 8, 0, 9)]
worst = 0.0
for N, c, order, n_ops, nsteps in shapes:
    if n_ops == 0:
        continue
    prob, ctrl, pcof, target = cases.synthetic_case(qgd, N=N, c=c, n_ops=n_ops, nsteps=nsteps, tf=0.02 * nsteps, seed=N + c)
    Gp, Gq, off = qgd.control_basis(ctrl, prob.nsteps, prob.tf, order // 2)
    ref = pp.evaluate(prob, Gp, Gq, off, pcof, target, order)
    dp = qgd.DeviceProblem(prob, order); dp.set_controls(ctrl); dp.set_target(target)
    grad, out3 = dp.discrete_adjoint(pcof)
    hist = np.zeros((2 * N, order // 2 + 1, nsteps + 1, c), order="F")
    qgd.eval_forward_(hist, prob, ctrl, pcof, order=order)
    href = pp.history_real(ref["ws"])
    eh = np.abs(hist - href).max() / np.abs(href).max()
    eg = np.abs(grad - ref["grad"]).max() / np.abs(ref["grad"]).max()
    worst = max(worst, eh, eg)
    print(f"N={N:3d} c={c:2d} order={order:2d} n_ops={n_ops} nsteps={nsteps:2d}: history {eh:.1e} gradient {eg:.1e}", flush=True)
    dp.close(); qgd.clear_cache()
print("worst", worst)
assert worst < 1e-11
