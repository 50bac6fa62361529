"""Config-5-shaped evaluation (N=256, 256 columns, 4 operators, order 12) against the number of time points: how much
of the 201-point evaluation is partial rounds of workgroups (a grid of 192 / 256 / 512 time points fills whole rounds).
Prints per-phase device time per time point."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa: F401
from __graft_entry__ import import_package
import cases

qgd = import_package()
steps = [int(a) for a in sys.argv[1:]] or [127, 191, 200, 255, 383, 511]
for S in steps:
    prob, ctrl, pcof, target = cases.synthetic_case(qgd, N=256, c=256, n_ops=4, nsteps=S, tf=2.0 * S / 200)
    target = (prob.u0 + 1j * prob.v0)
    dp = qgd.DeviceProblem(prob, 12)
    dp.set_controls(ctrl); dp.set_target(target); dp.set_timing(1)
    dp.discrete_adjoint(pcof)
    best = None
    for it in range(3):
        t0 = time.perf_counter(); dp.discrete_adjoint(pcof); t = time.perf_counter() - t0
        if best is None or t < best[0]:
            best = (t, dict(dp.timings()))
    t, ph = best
    nt = S + 1
    print(f"nt {nt:4d}: {t * 1e3:7.2f} ms  {t * 1e6 / nt:6.1f} us/point  " +
          "  ".join(f"{k} {v * 1e3 / nt:.1f}" for k, v in sorted(ph.items(), key=lambda kv: -kv[1])[:8]), flush=True)
    dp.close()
