"""Python wrapper against the bare C entry point, with-history call on cnot3 (diagnostic)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, ctypes as C
from __graft_entry__ import import_package
import bench
qgd = import_package()
prob, ctrl, pcof, target = bench.workload(qgd, 550, 550.0)
dp = qgd.DeviceProblem(prob, 8); dp.set_controls(ctrl); dp.set_target(target); dp.set_timing(0)
hist = dp.pin(np.zeros((128, 5, 551, 8), order="F")); lam = dp.pin(np.zeros((128, 5, 551, 8), order="F"))
forc = dp.pin(np.zeros((128, 551, 8), order="F"))
vp = lambda a: a.ctypes.data_as(C.c_void_p)
pc = np.ascontiguousarray(pcof); grad = np.zeros(len(pc)); out3 = np.zeros(3)
args = (dp.h, vp(pc), len(pc), 0, vp(grad), vp(hist), vp(lam), vp(forc), vp(out3))
def wrapper(): dp.discrete_adjoint(pcof, False, hist, lam, forc)
def bare(): dp.lib.qgd_discrete_adjoint(*args)
def bare_fresh():
    g = np.zeros(len(pc)); o = np.zeros(3)
    dp.lib.qgd_discrete_adjoint(dp.h, vp(pc), len(pc), 0, vp(g), vp(hist), vp(lam), vp(forc), vp(o))
for name, fn in (("wrapper", wrapper), ("bare", bare), ("bare_fresh", bare_fresh), ("wrapper", wrapper), ("bare", bare)):
    for _ in range(3): fn()
    ts = []
    for _ in range(20):
        t0 = time.perf_counter(); fn(); ts.append((time.perf_counter() - t0) * 1e3)
    print(f"{name:10s} mean {np.mean(ts):.3f} ms  min {np.min(ts):.3f}  max {np.max(ts):.3f}")
