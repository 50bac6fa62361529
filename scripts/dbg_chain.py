import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import torch, numpy as np
from __graft_entry__ import import_package
qgd = import_package()
import cases, proto_propagator as pp
order = 8
for nsteps in (550, 549, 540, 300):
    prob, ctrl, pcof, target = cases.cnot3_case(qgd, nsteps=nsteps, tf=float(nsteps))
    Gp, Gq, off = qgd.control_basis(ctrl, prob.nsteps, prob.tf, order // 2)
    ref = pp.evaluate(prob, Gp, Gq, off, pcof, target, order)
    shape = (128, 1 + order // 2, nsteps + 1, 8)
    hist = np.zeros(shape, order="F"); lam = np.zeros(shape, order="F"); forcing = np.zeros((128, nsteps + 1, 8), order="F")
    grad = np.zeros(len(pcof))
    qgd.discrete_adjoint_(grad, hist, lam, forcing, prob, ctrl, pcof, target, order=order)
    href = pp.history_real(ref["ws"])
    dh = np.abs(hist[:, 0] - href[:, 0]).max(axis=(0, 2))
    print(nsteps, "hist err max", dh.max(), "first bad n", (np.nonzero(dh > 1e-9)[0][:5]), "grad rel", np.abs(grad - ref["grad"]).max() / np.abs(ref["grad"]).max())
    lamref = np.asarray(ref["lam"])          # [nt, N, c] complex
    lamdev = lam[:64, 0] + 1j * lam[64:, 0]  # [N, nt, c]
    dl = np.abs(np.transpose(lamdev, (1, 0, 2)) - lamref).max(axis=(1, 2))
    print("   lam err max", dl.max(), "bad n range:", (lambda z: (z.min(), z.max(), len(z)) if len(z) else None)(np.nonzero(dl > 1e-9 * max(1, np.abs(lamref).max()))[0]))
    dp = qgd.device_problem(prob, order)
    print("   partition", dp.partition() if hasattr(dp, "partition") else "")
    qgd.clear_cache()
