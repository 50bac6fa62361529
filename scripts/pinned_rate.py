"""The reference-shaped call (three output arrays, 31.6 MB) into arrays pinned in different ways, on THIS box:
numpy arrays registered as they are (hipHostRegister: what INTEGRATION.md's shim does), the same over one DMA engine
(QGD_COPY_SPLIT=0), and pageable arrays; the raw D2H / H2D rates into runtime-allocated and into registered memory; the box's
NUMA layout and THP setting."""
import os, sys, time, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np, torch
    from __graft_entry__ import import_package
    import bench
    qgd = import_package()
    prob, ctrl, pcof, target = bench.workload(qgd, 550, 550.0)
    dp = qgd.DeviceProblem(prob, 8); dp.set_controls(ctrl); dp.set_target(target); dp.set_timing(0)
    for _ in range(10): dp.discrete_adjoint(pcof)
    mode = sys.argv[2]
    shape = (128, 5, 551, 8)
    def mk(s):
        a = np.zeros(s, order="F")
        if mode == "touched": a[...] = 1.0
        return a
    arrs = [mk(shape), mk(shape), mk((128, 551, 8))]
    if mode != "pageable":
        for a in arrs: dp.pin(a)
    for _ in range(5): dp.discrete_adjoint(pcof, False, *arrs)
    ts = []
    for _ in range(30):
        t0 = time.perf_counter(); dp.discrete_adjoint(pcof, False, *arrs); ts.append(time.perf_counter() - t0)
    print(json.dumps({"mode": mode, "copy_split": os.environ.get("QGD_COPY_SPLIT"), "median_ms": float(np.median(ts) * 1e3), "min_ms": float(np.min(ts) * 1e3)}))
    sys.exit(0)
def raw_rates():
    """D2H / H2D of 32 MB between the card and (a) memory the runtime allocates pinned (hipHostMalloc, through torch),
    (b) a numpy array registered in place (hipHostRegister, through the library's handle)"""
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np, torch
    n = 4 << 20
    dev = torch.zeros(n, dtype=torch.float64, device="cuda")
    res = {}
    pin = torch.empty(n, dtype=torch.float64).pin_memory()
    arr = np.zeros(n); arr[:] = 1.0
    reg = torch.from_numpy(arr)
    torch.cuda.cudart().cudaHostRegister(arr.ctypes.data, arr.nbytes, 0)
    for name, host in (("hipHostMalloc", pin), ("hipHostRegister(numpy)", reg)):
        for direction in ("d2h", "h2d"):
            ts = []
            for _ in range(8):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                if direction == "d2h": host.copy_(dev, non_blocking=True)
                else: dev.copy_(host, non_blocking=True)
                torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
            res[name + " " + direction] = round(n * 8 / min(ts) / 1e9, 1)
    torch.cuda.cudart().cudaHostUnregister(arr.ctypes.data)
    return res


try:
    st = open("/proc/self/status").read()
    print("process:", [l for l in st.splitlines() if l.startswith(("Cpus_allowed_list", "Mems_allowed_list"))])
    import glob
    print("gpu numa_node:", {p: open(p).read().strip() for p in glob.glob("/sys/class/drm/card*/device/numa_node")})
    print("nodes:", sorted(os.path.basename(p) for p in glob.glob("/sys/devices/system/node/node*")))
except Exception as e:
    print("numa info:", e)
if len(sys.argv) > 1 and sys.argv[1] == "--raw":
    print("GB/s:", raw_rates()); sys.exit(0)
out = subprocess.run([sys.executable, __file__, "--raw"], capture_output=True, text=True)
print([l for l in out.stdout.splitlines() if l.startswith("GB/s")] or out.stderr[-600:])
for f in ("enabled", "defrag", "shmem_enabled"):
    try: print("THP", f, open("/sys/kernel/mm/transparent_hugepage/" + f).read().strip())
    except Exception as e: print("THP", f, e)
for mode, env in (("pinned", None), ("pinned", "split0"), ("pageable", None), ("pinned", None)):
    e = dict(os.environ); e.pop("QGD_PIN_HUGEPAGE", None); e.pop("QGD_COPY_SPLIT", None)
    if env == "split0": e["QGD_COPY_SPLIT"] = "0"
    elif env: e["QGD_PIN_HUGEPAGE"] = env
    out = subprocess.run([sys.executable, __file__, "--child", mode], env=e, capture_output=True, text=True)
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    print(line[-1] if line else out.stderr[-400:], flush=True)
