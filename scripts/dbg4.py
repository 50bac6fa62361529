import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import torch
from __graft_entry__ import import_package, import_oracle
import cases
q=import_package(); o=import_oracle()
prob, ctrl, pcof, target = cases.cnot2_case(q)
order=4
rng=np.random.default_rng(8)
n2,c,nt = prob.real_system_size, prob.N_initial_conditions, prob.nsteps+1
term = rng.standard_normal((n2,c)); forcing = 0.1*rng.standard_normal((n2,nt,c))
for fo in (None, forcing):
    ref=o.eval_adjoint(prob,ctrl,pcof,term,order=order,forcing=fo)
    got=q.eval_adjoint(prob,ctrl,pcof,term,order=order,forcing=fo)
    d=np.abs(got[:,0]-ref[:,0]).max(axis=(0,2))
    print("forcing" if fo is not None else "no forcing", "err by n (first 6, last 6):", d[:6], d[-6:], "max ref", np.abs(ref[:,0]).max())
