"""The windowed entry points of round 4 at the scale they are for: cnot3 at order 2 with 55 000 steps
(examples/cnot3_optimize_gate.sb:27), resident (26.7 GiB) against a 2 GiB budget (14 windows): eval_adjoint with a random
terminal condition and forcing, the forced forward sweep (scalars and state history), the forced gradient against the adjoint
gradient, the gradient itself.    python3 scripts/windows_at_scale.py [nsteps] [budget_GiB]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from __graft_entry__ import import_package
import cases
qgd = import_package()
nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 55000
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
order = 2
prob, ctrl, pcof, target = cases.cnot3_case(qgd, nsteps=nsteps, tf=550.0)
rng = np.random.default_rng(0)
N2, c = prob.real_system_size, prob.N_initial_conditions
term = rng.standard_normal((N2, c))
forc = 1e-3 * rng.standard_normal((N2, nsteps + 1, c))
ff = np.asfortranarray(1e-3 * rng.standard_normal((N2, order // 2, nsteps + 1, c)))
res = {}
for label in ("resident", "windowed"):
    dp = qgd.DeviceProblem(prob, order)
    if label == "windowed":
        dp.set_memory_budget(int(budget * 2**30))
    print(label, dp.memory_plan(), flush=True)
    dp.set_controls(ctrl); dp.set_target(target)
    out = {}
    t0 = time.perf_counter(); out["grad"], out["o"] = dp.discrete_adjoint(pcof); t1 = time.perf_counter()
    out["lam"] = dp.eval_adjoint(pcof, term, forc)[:, 0].copy(); t2 = time.perf_counter()
    h = np.zeros((N2, 1 + order // 2, nsteps + 1, c), order="F")
    out["fs"] = np.asarray(dp.eval_forward_forced(pcof, ff, h)); out["fh"] = h[:, 0, ::97].copy(); t3 = time.perf_counter()
    del h
    out["gf"] = dp.eval_grad_forced(pcof); t4 = time.perf_counter()
    print(f"   first calls: gradient {t1 - t0:.2f} s, eval_adjoint {t2 - t1:.2f} s, forced sweep with history {t3 - t2:.2f} s, forced gradient {t4 - t3:.2f} s", flush=True)
    res[label] = out
    dp.close()
a, b = res["resident"], res["windowed"]
rel = lambda x, y: float(np.abs(np.asarray(x) - np.asarray(y)).max() / max(1e-300, np.abs(np.asarray(y)).max()))
print("gradient                 windowed vs resident:", rel(b["grad"], a["grad"]))
print("scalars                  windowed vs resident:", rel(b["o"], a["o"]))
print("eval_adjoint lambda      windowed vs resident:", rel(b["lam"], a["lam"]))
print("forced sweep scalars     windowed vs resident:", rel(b["fs"], a["fs"]))
print("forced sweep states      windowed vs resident:", rel(b["fh"], a["fh"]))
print("forced gradient          windowed vs resident:", rel(b["gf"], a["gf"]))
print("forced gradient vs adjoint gradient (resident):", rel(a["gf"], a["grad"]), " (windowed):", rel(b["gf"], b["grad"]))
