import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import torch, numpy as np
from __graft_entry__ import import_package
qgd = import_package()
import cases
for order in (2, 4):
    for name, prob, ctrl, pcof, target in cases.gradient_cases(qgd):
        res = {}
        for path in ("dense", "sparse"):
            dp = qgd.DeviceProblem(prob, order)
            try:
                dp.set_operator_path(path)
            except Exception as e:
                print(name, path, "unsupported"); dp.close(); continue
            dp.set_controls(ctrl); dp.set_target(target)
            grad, out3 = dp.discrete_adjoint(pcof)
            res[path] = (grad, dp.intermediate("L"), dp.intermediate("R"), dp.intermediate("sigma"), dp.operator_path())
            dp.close()
        if len(res) == 2:
            d, s = res["dense"], res["sparse"]
            print(name, order, "N", prob.N_tot_levels, s[4], "dL", np.abs(d[1]-s[1]).max(), "dR", np.abs(d[2]-s[2]).max(),
                  "dsigma", np.abs(d[3]-s[3]).max(), "of", np.abs(d[3]).max(), "dgrad", np.abs(d[0]-s[0]).max())
