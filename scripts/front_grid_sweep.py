"""Fused front against the general path over grid sizes around every boundary of the launch logic (one round / several rounds of
workgroups, with / without pre-built panels, the two instantiations of the elimination): gradient and scalars, cnot3 order 8.
   gpurun -- python scripts/front_grid_sweep.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
qgd = ge.import_package()
import cases
worst = 0.0
for nt in (2, 3, 9, 255, 256, 257, 258, 511, 512, 513, 514, 551, 600, 703, 704, 705, 706, 767, 768, 769, 1001):
    prob, ctrl, pcof, target = cases.cnot3_case(qgd, nsteps=nt - 1, tf=float(nt - 1))
    res = {}
    for tag, paths in (("front", "front"), ("general", "no_front")):
        os.environ["QGD_PATHS"] = paths
        dp = qgd.DeviceProblem(prob, 8); dp.set_target(target); dp.set_controls(ctrl)
        g, o = dp.discrete_adjoint(pcof); g2, o2 = dp.discrete_adjoint(pcof)
        res[tag] = (g, np.asarray(o), dp.front_path_taken(), np.array_equal(g, g2))
        dp.close()
    f, g = res["front"], res["general"]
    err = np.abs(f[0] - g[0]).max() / np.abs(g[0]).max()
    worst = max(worst, err, np.abs(f[1] - g[1]).max())
    print(f"nt {nt:5d}: front {f[2]} / {g[2]}  grad rel diff {err:.2e}  scalars {np.abs(f[1] - g[1]).max():.2e}  repeatable {f[3]}", flush=True)
    assert f[2] and not g[2] and f[3]
print("worst", worst)
sys.exit(0 if worst < 1e-11 else 1)
