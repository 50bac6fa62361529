"""Dispersive multi-qubit problems (2 levels each, sparse operators) with 2 .. 6 subsystems = control pairs: N = 4 .. 64.
Five and six controls take the generic (run-time operator count) instantiations of the sparse kernels."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from __graft_entry__ import import_package
qgd = import_package()
for q in (2, 3, 4, 5, 6):
    for order in (8, 12):
        sizes = (2,) * q
        freqs = np.linspace(4.1, 4.9, q)
        kerr = np.zeros((q, q))
        for i in range(q):
            kerr[i, i] = -0.2
            for j in range(i + 1, q): kerr[i, j] = kerr[j, i] = -0.005
        nsteps, tf = 400, 400.0
        prob = qgd.DispersiveProblem(sizes, sizes, freqs, freqs, kerr, tf, nsteps)
        ctrl = [qgd.FortranBSplineControl(2, 12, tf) for _ in range(q)]
        pcof = (np.random.default_rng(0).random(qgd.get_number_of_control_parameters(ctrl)) - 0.5) * 0.01
        N = prob.N_tot_levels
        target = np.eye(N, prob.N_initial_conditions).astype(complex)
        dp = qgd.DeviceProblem(prob, order); dp.set_controls(ctrl); dp.set_target(target)
        dp.set_timing(1)
        for _ in range(3): dp.discrete_adjoint(pcof)
        tm = dp.timings()
        path = dp.operator_path()
        dp.set_timing(0)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): dp.discrete_adjoint(pcof)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        top = sorted(tm.items(), key=lambda kv: -kv[1])[:4]
        print(f"{q} qubits N={N:3d} c={prob.N_initial_conditions:3d} order {order:2d} path {path}: {dt*1e3:7.3f} ms   " + "  ".join(f"{k} {v:.3f}" for k, v in top), flush=True)
        dp.close(); qgd.clear_cache()
