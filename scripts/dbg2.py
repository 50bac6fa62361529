import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from __graft_entry__ import import_package, import_oracle
import cases, proto_propagator as pp
q=import_package(); o=import_oracle()
prob, ctrl, pcof, target = cases.cnot3_case(q)
order=8; m=4
Gp,Gq,off = q.control_basis(ctrl, prob.nsteps, prob.tf, m)
ref = pp.evaluate(prob, Gp,Gq,off,pcof,target,order)
dp = q.device_problem(prob, order); dp.set_controls(ctrl); dp.set_target(target)
shape=(128,5,21,8)
hist=np.zeros(shape,order='F'); lam=np.zeros(shape,order='F'); forc=np.zeros((128,21,8),order='F')
grad,out3 = dp.discrete_adjoint(pcof, False, hist, lam, forc)
lamr = np.concatenate([ref['lam'].real, ref['lam'].imag],axis=1)  # nt,2N,c
fr = np.concatenate([ref['f'].real, ref['f'].imag],axis=1)
print("lam diff", np.abs(np.transpose(lamr,(1,0,2))-lam[:,0]).max(), "forcing diff", np.abs(np.transpose(fr,(1,0,2))-forc).max())
H = pp.history_real(ref['ws']); print("hist diff", np.abs(H-hist).max())
sig = dp.intermediate("sigma")
print("sigP diff", np.abs(sig[...,0]-ref['sigP']).max(), "sigQ diff", np.abs(sig[...,1]-ref['sigQ']).max(), np.abs(ref['sigP']).max(), np.abs(ref['sigQ']).max())
d = np.abs(sig[...,0]-ref['sigP']); print("argmax sigP", np.unravel_index(d.argmax(), d.shape))
d = np.abs(sig[...,1]-ref['sigQ']); print("argmax sigQ", np.unravel_index(d.argmax(), d.shape))
print("grad rel", np.abs(grad-ref['grad']).max()/np.abs(ref['grad']).max())
gi = np.abs(grad-ref['grad']); print("worst grad idx", gi.argmax(), gi.max())
# per-time-point error profile
print("sigQ err by n", np.abs(sig[...,1]-ref['sigQ']).max(axis=(1,2)))
print("sigP err by d", np.abs(sig[...,0]-ref['sigP']).max(axis=(0,1)), "sigQ err by d", np.abs(sig[...,1]-ref['sigQ']).max(axis=(0,1)))
print("by op", np.abs(sig[...,0]-ref['sigP']).max(axis=(0,2)), np.abs(sig[...,1]-ref['sigQ']).max(axis=(0,2)))
