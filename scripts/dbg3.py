import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import torch
from __graft_entry__ import import_package
import cases, proto_propagator as pp
q=import_package()
for order in (2,4,6,10,12):
    prob, ctrl, pcof, target = cases.cnot3_case(q, nsteps=30, tf=15.0)
    Gp,Gq,off = q.control_basis(ctrl, prob.nsteps, prob.tf, order//2)
    ref = pp.evaluate(prob, Gp,Gq,off,pcof,target,order)
    shape=(128,1+order//2,31,8)
    hist=np.zeros(shape,order='F'); lam=np.zeros(shape,order='F'); forc=np.zeros((128,31,8),order='F'); grad=np.zeros(len(pcof))
    q.discrete_adjoint_(grad,hist,lam,forc,prob,ctrl,pcof,target,order=order)
    H=pp.history_real(ref['ws'])
    print(order, "hist err %.2e (max %.2e)"%(np.abs(hist-H).max(), np.abs(H).max()), "grad rel %.2e"%(np.abs(grad-ref['grad']).max()/np.abs(ref['grad']).max()))
    for j in range(1+order//2): print("   j",j,"%.2e"%np.abs(hist[:,j]-H[:,j]).max())
    g2=q.discrete_adjoint(prob,ctrl,pcof,target,order=order)
    print("   grad2 rel %.2e"%(np.abs(g2-ref['grad']).max()/np.abs(ref['grad']).max()))
    q.clear_cache()
