import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from __graft_entry__ import import_package, import_oracle
import cases
q=import_package(); o=import_oracle()
prob, ctrl, pcof, target = cases.cnot3_case(q)
for tol in (1e-14, 1e-15):
    prob.gmres_abstol=prob.gmres_reltol=tol
    g_ref,h_ref,l_ref,f_ref,st = o.discrete_adjoint(prob, ctrl, pcof, target, order=8, return_all=True)
    dp = q.device_problem(prob, 8); dp.set_controls(ctrl); dp.set_target(target)
    hist = np.zeros(h_ref.shape, order='F'); out3 = dp.eval_forward(pcof, hist)
    for j in range(5):
        print(tol, "j",j,"max diff %.3e  max ref %.3e"%(np.abs(hist[:,j]-h_ref[:,j]).max(), np.abs(h_ref[:,j]).max()))
    grad,_ = dp.discrete_adjoint(pcof, True)
    print("grad rel", np.abs(grad-g_ref).max()/np.abs(g_ref).max(), "iters", st.fwd_gmres_iters, st.adj_gmres_iters)
    print(dp.timings())
# full size timing
prob, target = q.cnot3_problem(nsteps=550, tf=550.0)
ctrl = cases.cnot3_controls(q, prob)
pcof = (np.random.default_rng(0).random(180) - 0.5) * 2 * np.pi * 0.005
dp = q.device_problem(prob, 8); dp.set_controls(ctrl); dp.set_target(target)
import time
for it in range(3):
    t=time.time(); g,o3 = dp.discrete_adjoint(pcof); dt=time.time()-t
    print("full eval wall %.2f ms"%(dt*1e3), {k:round(v,3) for k,v in dp.timings().items()})
hist = np.zeros((128,5,551,8),order='F'); dp.eval_forward(pcof,hist)
psi = hist[:64,0,-1]+1j*hist[64:,0,-1]; print("gram dev dt=1", np.abs(psi.conj().T@psi-np.eye(8)).max())
prob2, _ = q.cnot3_problem(nsteps=1100, tf=550.0)
ctrl2 = cases.cnot3_controls(q, prob2)
dp2 = q.device_problem(prob2, 8); dp2.set_controls(ctrl2)
hist2 = np.zeros((128,5,1101,8),order='F'); dp2.eval_forward(pcof,hist2)
psi2 = hist2[:64,0,-1]+1j*hist2[64:,0,-1]; print("gram dev dt=.5", np.abs(psi2.conj().T@psi2-np.eye(8)).max(), "state diff", np.abs(psi-psi2).max())
