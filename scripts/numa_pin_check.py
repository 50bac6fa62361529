"""Reproduces the slow box on any box: the process's memory policy is bound to the NUMA node the card is NOT attached to
(set_mempolicy), the three output arrays of the reference-shaped call are created there, pinned, and the call is timed -- with
the library moving the pages to the card's node at registration (QGD_PIN_NUMA=1) and without (default)."""
import os, sys, time, json, subprocess, ctypes, glob
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    far = int(sys.argv[2])
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np, torch
    from __graft_entry__ import import_package
    import bench
    qgd = import_package()
    prob, ctrl, pcof, target = bench.workload(qgd, 550, 550.0)
    dp = qgd.DeviceProblem(prob, 8); dp.set_controls(ctrl); dp.set_target(target); dp.set_timing(0)
    for _ in range(10): dp.discrete_adjoint(pcof)
    if far >= 0:
        libc = ctypes.CDLL(None, use_errno=True)
        mask = ctypes.c_ulong(1 << far)
        rc = libc.syscall(238, 2, ctypes.byref(mask), 65)          # set_mempolicy(MPOL_BIND, far node)
        assert rc == 0, ctypes.get_errno()
    shape = (128, 5, 551, 8)
    arrs = [np.zeros(shape, order="F"), np.zeros(shape, order="F"), np.zeros((128, 551, 8), order="F")]
    for a in arrs: a[...] = 1.0                                     # (touched: the pages exist, on the far node)
    if far >= 0:
        libc.syscall(238, 0, None, 0)                               # back to the default policy
    for a in arrs: dp.pin(a)
    for _ in range(5): g, o = dp.discrete_adjoint(pcof, False, *arrs)
    ts = []
    for _ in range(30):
        t0 = time.perf_counter(); dp.discrete_adjoint(pcof, False, *arrs); ts.append(time.perf_counter() - t0)
    nodes = {}
    for line in open("/proc/self/numa_maps"):
        if format(arrs[0].ctypes.data & ~0xfff, "x") in line.split()[0] or format((arrs[0].ctypes.data & ~0x1fffff), "x") == line.split()[0]:
            nodes = {k: v for k, v in (t.split("=") for t in line.split() if t.startswith("N"))}
    print(json.dumps({"far_node": far, "QGD_PIN_NUMA": os.environ.get("QGD_PIN_NUMA"), "median_ms": float(np.median(ts) * 1e3), "pages_of_first_array_by_node": nodes,
                      "hist_checksum": float(np.abs(arrs[0]).sum())}))
    sys.exit(0)
import torch
p = torch.cuda.get_device_properties(0)
bdf = "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
node = int(open(f"/sys/bus/pci/devices/{bdf}/numa_node").read())
nn = len(glob.glob("/sys/devices/system/node/node[0-9]*"))
print(f"card {bdf} on NUMA node {node} of {nn}", flush=True)
far = (node + 1) % nn if nn > 1 and node >= 0 else -1
for f, env in ((-1, None), (far, None), (far, "1"), (far, None), (far, "1")):
    e = dict(os.environ); e.pop("QGD_PIN_NUMA", None)
    if env is not None: e["QGD_PIN_NUMA"] = env
    out = subprocess.run([sys.executable, __file__, "--child", str(f)], env=e, capture_output=True, text=True)
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    print(line[-1] if line else out.stderr[-600:], flush=True)
