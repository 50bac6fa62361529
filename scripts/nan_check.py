import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
from __graft_entry__ import import_package
import cases
qgd = import_package()
for name, case, order in (("cnot2", cases.cnot2_case(qgd, nsteps=40, tf=40.0), 8), ("cnot3", cases.cnot3_case(qgd, nsteps=60, tf=60.0), 8)):
    prob, ctrl, pcof, target = case
    for small in (True, False):
        dp = qgd.DeviceProblem(prob, order); dp.set_small_path(small); dp.set_controls(ctrl); dp.set_target(target)
        bad = pcof.copy(); bad[3] = np.nan
        try:
            g, o = dp.discrete_adjoint(bad)
            print(name, "small" if small else "general", "taken", dp.small_path_taken(), "-> finite grad:", np.isfinite(g).all(), "out", o)
        except qgd._lib.QGDError as e:
            print(name, "small" if small else "general", "-> error", e.code, str(e)[:80])
        g, o = dp.discrete_adjoint(pcof)          # the handle is usable afterwards
        print("   after:", np.isfinite(g).all(), o)
        dp.close()
