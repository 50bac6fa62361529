"""cnot3 headline evaluation (N=64, 8 columns, order 8, 550 steps) under the switches of the front pipeline:
QGD_PIPE_CHUNKS (time chunks on side streams), QGD_INV_MULTI (aligned multi-matrix inverse), QGD_INV_PIVOTED (no
static-pivot attempt), QGD_INV_OLD (round-1 kernel k_inverse_mfma).
One subprocess per setting (the switches are read when the handle is created / per launch)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, time, os
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np, torch
from __graft_entry__ import import_package
import bench
qgd = import_package()
prob, ctrl, pcof, target = bench.workload(qgd, 550, 550.0)
dp = qgd.DeviceProblem(prob, 8); dp.set_controls(ctrl); dp.set_target(target); dp.set_timing(0)
for _ in range(5): g, o = dp.discrete_adjoint(pcof)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(40): g, o = dp.discrete_adjoint(pcof)
torch.cuda.synchronize(); el = (time.perf_counter() - t0) / 40
print("%%-40s %%.1f us  grad_norm %%.12f" %% (os.environ.get("TAG"), el * 1e6, np.linalg.norm(g)))
''' % (ROOT, ROOT)
settings = [dict(), dict(QGD_GRAPH="1"), dict(QGD_INV_STATIC="1"), dict(QGD_INV_STATIC="1", QGD_INV_PIVOTED="1"), dict(QGD_INV_MULTI="3"), dict(QGD_INV_MULTI="2"),
            dict(QGD_PIPE_CHUNKS="2"), dict(QGD_PIPE_CHUNKS="3"), dict(), dict(QGD_GRAPH="1")]
if len(sys.argv) > 1:      # settings from the command line: "default" or "K=V,K=V" per argument
    settings = [dict() if a == "default" else dict(kv.split("=", 1) for kv in a.split(",")) for a in sys.argv[1:]]
for extra in settings:
    env = dict(os.environ, TAG=" ".join(f"{k}={v}" for k, v in extra.items()) or "default", **extra)
    subprocess.run([sys.executable, "-c", CODE], env=env)
