"""Per-phase cycle profile of the sparse kernels.  Build the library with the hooks first:
    make -C quantumgatedesign.jl_amd/csrc CXXFLAGS="-O3 -std=c++17 -fPIC -DQGD_SPARSE_PROFILE"
(and rebuild without the flag afterwards)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch, numpy as np
from __graft_entry__ import import_package
qgd = import_package()
prob, target = qgd.cnot3_problem(nsteps=550, tf=550.0)
ctrl = [qgd.CarrierControl(qgd.FortranBSplineControl(2, 10, prob.tf), [0.0, -1.38, -6e-6]) for _ in range(3)]
pcof = 0.01 * np.random.default_rng(0).standard_normal(qgd.get_number_of_control_parameters(ctrl))
dp = qgd.DeviceProblem(prob, 8); dp.set_controls(ctrl); dp.set_target(target)
lib = qgd._lib.lib()
out = (C.c_ulonglong * 32)()
for _ in range(3): dp.discrete_adjoint(pcof)
lib.qgdk_sparse_profile(out, 1)
K = 10
for _ in range(K): dp.discrete_adjoint(pcof)
lib.qgdk_sparse_profile(out, 0)
names = {0: "build: assemble", 1: "build: ecol+sync", 2: "build: source 0", 3: "build: levels", 4: "build: output",
         16: "grad: assemble", 17: "grad: lists+seeds", 18: "grad: G passes", 19: "grad: D parts (sum)", 20: "grad: S parts (sum)",
         21: "grad: (loop tail)", 22: "grad: reduce+out"}
for i, nm in names.items():
    print(f"{nm:24s} {out[i] / K:10.0f} cycles")
