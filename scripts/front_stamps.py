"""In-kernel wall-clock stamps of k_tables_front and k_psi (library built with -DQGD_STAMPS: bash scripts/build_variant.sh stamps -DQGD_STAMPS).
   gpurun -- env QGD_LIB_PATH=scripts/ubench/bin/libqgd_stamps.so python scripts/front_stamps.py"""
import os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
qgd = ge.import_package()
import bench
prob, ctrl, pcof, target = bench.workload(qgd, 550, 550.0)
dp = qgd.DeviceProblem(prob, 8); dp.set_controls(ctrl); dp.set_target(target)
for _ in range(20): dp.discrete_adjoint(pcof)
L = qgd._lib.lib()
b = np.zeros((512, 8), dtype=np.uint64); c = np.zeros((1024, 8), dtype=np.uint64)
L.qgdk_stamps_build(b.ctypes.data_as(C.c_void_p)); L.qgdk_stamps_chain(c.ctypes.data_as(C.c_void_p))
b = b.astype(np.int64); c = c.astype(np.int64)
npre = 78
t0 = b[:256, 0].min()
print("k_tables_front, us since the first workgroup's start (pre-building workgroups: start, tables done, build done, [phi0 start, done]):")
for j in (0, 1, 40, 78):
    print("  pre wg", j, np.round((b[j, :3] - t0) * 0.01, 2))
pre = b[:npre]
print("  pre: mean start %.2f tables %.2f build-end %.2f; latest build-end %.2f" % (((pre[:, 0] - t0).mean() * .01), ((pre[:, 1] - t0).mean() * .01), ((pre[:, 2] - t0).mean() * .01), ((pre[:, 2] - t0).max() * .01)))
tb = b[npre:256]
print("  table wgs: mean start %.2f end %.2f latest end %.2f" % (((tb[:, 0] - t0).mean() * .01), ((tb[:, 1] - t0).mean() * .01), ((tb[:, 1] - t0).max() * .01)))
t0 = c[:551, 0].min()
d = (c[:551, :4] - t0) * 0.01
print("k_psi (per time point: start, operands in registers, psi done, h done), us: mean", np.round(d.mean(axis=0), 2), "max", np.round(d.max(axis=0), 2))
for n in (0, 1, 100, 300, 549, 550):
    print("  n", n, np.round(d[n], 2))
print("  k_psi workgroups that start later than 2 us:", int((d[:, 0] > 2).sum()))
