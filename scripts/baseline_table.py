"""The rows of BASELINE.md section 4 that one GPU can fill: C2 (cnot2, order 8), C3 (cnot3, order 8), C5 -- wall time of
one full gradient evaluation (median of 7 after 3 warm-ups, host to host), and the actual (not just within-tolerance)
errors against the CPU oracle on the same problems at a reduced number of steps (GMRES 1e-15, converged terminal solve)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from __graft_entry__ import import_package, import_oracle
import cases, bench
qgd = import_package(); orc = import_oracle(); orc.lib()


def median_eval(dp, pcof, n=7):
    for _ in range(3): dp.discrete_adjoint(pcof)
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); dp.discrete_adjoint(pcof); torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts))


def errors(case, order):
    prob, ctrl, pcof, target = case
    orc.set_converged_terminal(True)
    try:
        g_ref, h_ref, lam_ref, f_ref, st = orc.discrete_adjoint(prob, ctrl, pcof, target, order=order, return_all=True)
    finally:
        orc.set_converged_terminal(False)
    hist = np.zeros(h_ref.shape, order="F"); grad = np.zeros_like(g_ref)
    qgd.discrete_adjoint_(grad, hist, None, None, prob, ctrl, pcof, target, order=order)
    qgd.clear_cache()
    return np.abs(grad - g_ref).max() / np.abs(g_ref).max(), np.abs(hist - h_ref).max()


# C2: examples/cnot2_optimization.jl:10-47 (tf = 100, nsteps = 100), order 8
prob, ctrl, pcof, target = cases.cnot2_case(qgd, nsteps=100, tf=100.0, amp=1e-2)
dp = qgd.DeviceProblem(prob, 8); dp.set_controls(ctrl); dp.set_target(target); dp.set_timing(0)
t2 = median_eval(dp, pcof); dp.close()
eg, eh = errors(cases.cnot2_case(qgd, nsteps=100, tf=100.0, amp=1e-2), 8)
print(f"C2 cnot2 order 8, 100 steps: T_eval {t2 * 1e6:.1f} us, {100 / t2:.4g} timesteps/s; vs oracle (full size): gradient rel err {eg:.1e}, history abs err {eh:.1e}")
# C3: cnot3 headline
prob, ctrl, pcof, target = bench.workload(qgd, 550, 550.0)
dp = qgd.DeviceProblem(prob, 8); dp.set_controls(ctrl); dp.set_target(target); dp.set_timing(0)
t3 = median_eval(dp, pcof); dp.close()
eg, eh = errors(cases.cnot3_case(qgd, nsteps=20, tf=20.0), 8)
print(f"C3 cnot3 order 8, 550 steps: T_eval {t3 * 1e6:.1f} us, {550 / t3:.4g} timesteps/s; vs oracle (20 steps at dt = 1): gradient rel err {eg:.1e}, history abs err {eh:.1e}")
print(f"   history traffic B_step*nsteps/T_eval = {98e3 * 550 / t3 / 1e9:.0f} GB/s = {98e3 * 550 / t3 / 8e12:.3f} of 8 TB/s")
# C5
prob, ctrl, pcof, _ = cases.synthetic_case(qgd, N=256, c=256, n_ops=4, nsteps=200, tf=2.0)
target = prob.u0 + 1j * prob.v0
dp = qgd.DeviceProblem(prob, 12); dp.set_controls(ctrl); dp.set_target(target); dp.set_timing(0)
t5 = median_eval(dp, pcof, 5); dp.close()
print(f"C5 synthetic N=256, 256 columns, order 12, 200 steps: T_eval {t5 * 1e3:.2f} ms, {200 / t5:.4g} timesteps/s; "
      f"history traffic {16.8e6 * 200 / t5 / 1e9:.0f} GB/s = {16.8e6 * 200 / t5 / 8e12:.3f} of 8 TB/s")
