"""cnot3 (N = 64, sparse operators, order 8, 550 steps) against the number of initial-condition columns: looks for
performance cliffs in the column dimension.  python scripts/c_sweep_timing.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from __graft_entry__ import import_package
import cases
qgd = import_package()
for cols in (1, 4, 8, 9, 16, 24, 32, 48, 64):
    prob, ctrl, pcof, target = cases.cnot3_case(qgd, nsteps=550, tf=550.0)
    rng = np.random.default_rng(9)
    z = rng.standard_normal((prob.N_tot_levels, cols)) + 1j * rng.standard_normal((prob.N_tot_levels, cols))
    z /= np.linalg.norm(z, axis=0)
    prob.u0, prob.v0 = np.asfortranarray(z.real), np.asfortranarray(z.imag)
    prob.N_initial_conditions = cols
    target = rng.standard_normal((prob.N_tot_levels, cols)) + 1j * rng.standard_normal((prob.N_tot_levels, cols))
    dp = qgd.DeviceProblem(prob, 8); dp.set_controls(ctrl); dp.set_target(target)
    dp.set_timing(1)
    for _ in range(3): dp.discrete_adjoint(pcof)
    tm = dp.timings()
    dp.set_timing(0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): dp.discrete_adjoint(pcof)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    top = sorted(tm.items(), key=lambda kv: -kv[1])[:5]
    print(f"c={cols:3d}: {dt*1e3:7.3f} ms   " + "  ".join(f"{k} {v:.3f}" for k, v in top), flush=True)
    dp.close(); qgd.clear_cache()
