"""cnot3 evaluation time against the scan geometry (QGD_SCAN_B0 blocks, QGD_SCAN_B2 super-blocks): plain loop, us."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "one":
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np, torch
    from __graft_entry__ import import_package
    import bench
    qgd = import_package()
    prob, ctrl, pcof, target = bench.workload(qgd, 550, 550.0)
    dp = qgd.DeviceProblem(prob, 8); dp.set_controls(ctrl); dp.set_target(target); dp.set_timing(0)
    g0, _ = dp.discrete_adjoint(pcof)
    best = 1e9
    for rep in range(3):
        for _ in range(5): dp.discrete_adjoint(pcof)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(40): dp.discrete_adjoint(pcof)
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 40 * 1e6)
    print(f"B0={os.environ.get('QGD_SCAN_B0','-'):>3} B2={os.environ.get('QGD_SCAN_B2','-'):>3}  {best:7.1f} us  |grad|={np.linalg.norm(g0):.12e}")
else:
    for b0, b2 in [(None, None), (64, 8), (64, 13), (60, None), (58, None), (56, None), (55, None), (54, None), (52, None), (50, None), (56, 8), (56, 9), (56, 12), (56, 14), (55, 8), (55, 12), (None, None)]:
        env = dict(os.environ)
        if b0: env["QGD_SCAN_B0"] = str(b0)
        if b2: env["QGD_SCAN_B2"] = str(b2)
        subprocess.call([sys.executable, os.path.abspath(__file__), "one"], env=env)
