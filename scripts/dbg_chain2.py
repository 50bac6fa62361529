import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import torch, numpy as np
from __graft_entry__ import import_package
qgd = import_package()
import cases, proto_propagator as pp
order = 4
for which, nsteps in (("cnot2", 300), ("cnot2", 550), ("guarded", 300)):
    prob, ctrl, pcof, target = getattr(cases, which + "_case")(qgd, nsteps=nsteps, tf=nsteps * 0.5)
    Gp, Gq, off = qgd.control_basis(ctrl, prob.nsteps, prob.tf, order // 2)
    ref = pp.evaluate(prob, Gp, Gq, off, pcof, target, order)
    grad = qgd.discrete_adjoint(prob, ctrl, pcof, target, order=order)
    print(which, nsteps, "N", prob.N_tot_levels, "grad rel", np.abs(grad - ref["grad"]).max() / np.abs(ref["grad"]).max())
    qgd.clear_cache()
