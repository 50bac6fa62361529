"""Where does the fused front win?  cnot3, order 8, us per evaluation (median of 4 x 100) of QGD_PATHS=front against no_front over grid sizes,
both handles alive in one process, interleaved.  (qgd_host_eval.cpp: front_applies takes the front by itself on 513 .. 704 time points.)
   gpurun -- python scripts/front_crossover.py"""
import os, sys, time
import numpy as np
ROOT = "/root/repo"
ROOT = os.environ.get("GRAFT_REPO_ROOT", ROOT)
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
qgd = ge.import_package()
import cases
import sys as _s
GRID = [int(a) for a in _s.argv[2:]] or (20, 60, 100, 150, 200, 256, 300, 400, 512, 550, 700, 800, 1100)
FRONT = _s.argv[1] if len(_s.argv) > 1 else 'front'
for ns in GRID:
    prob, ctrl, pcof, target = cases.cnot3_case(qgd, nsteps=ns, tf=float(ns))
    out = {}
    dps = {}
    for tag, paths in (("front", FRONT), ("general", "no_front")):
        os.environ["QGD_PATHS"] = paths
        dp = qgd.DeviceProblem(prob, 8); dp.set_target(target); dp.set_controls(ctrl)
        for _ in range(30): dp.discrete_adjoint(pcof)
        dps[tag] = dp
    ts = {"front": [], "general": []}
    for rep in range(4):
        for tag in ("front", "general"):
            os.environ["QGD_PATHS"] = FRONT if tag == "front" else "no_front"
            dp = dps[tag]
            t0 = time.perf_counter()
            for _ in range(100): dp.discrete_adjoint(pcof)
            ts[tag].append((time.perf_counter() - t0) / 100)
    print(f"nsteps {ns:5d}: front {np.median(ts['front'])*1e6:7.1f} us  general {np.median(ts['general'])*1e6:7.1f} us", flush=True)
    for dp in dps.values(): dp.close()
