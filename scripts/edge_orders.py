import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
from __graft_entry__ import import_package
import cases, proto_propagator as pp
qgd = import_package()
for which, order in (("cnot2", 18), ("cnot2", 24), ("cnot3", 20), ("synthetic", 18), ("synthetic", 24)):
    try:
        if which == "synthetic":
            prob, ctrl, pcof, target = cases.synthetic_case(qgd, N=80, c=8, n_ops=2, nsteps=8, tf=0.08)
        else:
            prob, ctrl, pcof, target = getattr(cases, which + "_case")(qgd, nsteps=12, tf=6.0)
        dp = qgd.DeviceProblem(prob, order); dp.set_controls(ctrl); dp.set_target(target)
        g, o = dp.discrete_adjoint(pcof)
        Gp, Gq, off = qgd.control_basis(ctrl, prob.nsteps, prob.tf, order // 2)
        ref = pp.evaluate(prob, Gp, Gq, off, pcof, target, order)
        print(which, order, "ok  grad rel err", np.abs(g - ref["grad"]).max() / np.abs(ref["grad"]).max(), dp.operator_path())
        dp.close()
    except Exception as e:
        print(which, order, "->", type(e).__name__, str(e)[:150])
