"""The reference-shaped call discrete_adjoint!(grad, history, lambda_history, adjoint_forcing, ...) on the cnot3 headline
problem with the three output arrays registered: ms per evaluation (device staging + copies; the
zero-copy writer it once compared with was slower and is gone)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from __graft_entry__ import import_package
import bench
qgd = import_package()
prob, ctrl, pcof, target = bench.workload(qgd, 550, 550.0)
dp = qgd.DeviceProblem(prob, 8); dp.set_controls(ctrl); dp.set_target(target); dp.set_timing(0)
g0, _ = dp.discrete_adjoint(pcof)
hist = dp.pin(np.zeros((128, 5, 551, 8), order="F")); lam = dp.pin(np.zeros((128, 5, 551, 8), order="F"))
forc = dp.pin(np.zeros((128, 551, 8), order="F"))
for _ in range(3): dp.discrete_adjoint(pcof, False, hist, lam, forc)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): g, _ = dp.discrete_adjoint(pcof, False, hist, lam, forc)
torch.cuda.synchronize(); el = (time.perf_counter() - t0) / 20
h2 = np.zeros((128, 5, 551, 8), order="F"); dp.eval_forward(pcof, h2)
print("%.3f ms per evaluation with history; grad diff %.1e, history diff %.1e" %
      (el * 1e3, np.abs(g - g0).max(), np.abs(h2 - hist).max()))
# where the time of one call goes: the bare C entry point against the Python wrapper around it
import ctypes as C
vp = lambda a: a.ctypes.data_as(C.c_void_p)
pc = np.ascontiguousarray(pcof); grad = np.zeros(len(pc)); out3 = np.zeros(3)
args = (dp.h, vp(pc), len(pc), 0, vp(grad), vp(hist), vp(lam), vp(forc), vp(out3))
for _ in range(3): dp.lib.qgd_discrete_adjoint(*args)
t0 = time.perf_counter()
for _ in range(20): dp.lib.qgd_discrete_adjoint(*args)
print("bare qgd_discrete_adjoint with the three arrays: %.3f ms per call" % ((time.perf_counter() - t0) / 20 * 1e3))
