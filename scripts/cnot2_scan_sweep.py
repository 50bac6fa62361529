"""cnot2 (N=4, order 8, 100 steps): evaluation time against the scan geometry (QGD_SCAN_B0 blocks; <= 8 blocks: no second
scan level, two launches fewer).  One process per setting."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "one":
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np, torch
    from __graft_entry__ import import_package
    import cases
    qgd = import_package()
    prob, ctrl, pcof, target = cases.cnot2_case(qgd, nsteps=100, tf=100.0, amp=1e-2)
    dp = qgd.DeviceProblem(prob, 8); dp.set_controls(ctrl); dp.set_target(target)
    dp.set_timing(1)
    for _ in range(3): g0, _ = dp.discrete_adjoint(pcof)
    tm = dp.timings(); dp.set_timing(0)
    best = 1e9
    for rep in range(3):
        for _ in range(10): dp.discrete_adjoint(pcof)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(100): dp.discrete_adjoint(pcof)
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 100 * 1e6)
    print(f"B0={os.environ.get('QGD_SCAN_B0','-'):>3} {best:7.1f} us |grad|={np.linalg.norm(g0):.12e}", {k: round(v * 1e3, 1) for k, v in tm.items()})
else:
    for b0 in (None, 4, 5, 6, 7, 8, 10, 12):
        env = dict(os.environ)
        if b0: env["QGD_SCAN_B0"] = str(b0)
        subprocess.call([sys.executable, os.path.abspath(__file__), "one"], env=env)
