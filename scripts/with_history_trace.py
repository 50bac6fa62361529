"""The reference-shaped call as bench.py measures it, with the wall time of every call, run as the FIRST process on a box
(debugging aid for the 0.89 / 3 ms spread of `with_history.pinned_ms_per_step` between otherwise identical runs)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from __graft_entry__ import import_package
import bench
qgd = import_package()
prob, ctrl, pcof, target = bench.workload(qgd, 550, 550.0)
dp = qgd.DeviceProblem(prob, 8); dp.set_controls(ctrl); dp.set_target(target); dp.set_timing(0)
pre = int(sys.argv[1]) if len(sys.argv) > 1 else 285
for _ in range(pre): dp.discrete_adjoint(pcof)
for _ in range(20): dp.eval_forward(pcof)
torch.cuda.synchronize()


def nodes_of(a):
    lo = a.ctypes.data
    out = {}
    for line in open("/proc/self/numa_maps"):
        f = line.split()
        addr = int(f[0], 16)
        if addr <= lo < addr + (64 << 20):
            cand = {k: v for k, v in (t.split("=") for t in f[1:] if t[0] == "N" and "=" in t)}
            if cand: out = cand
    return out


import gc
GC_OFF = len(sys.argv) > 2 and sys.argv[2] == "nogc"
gc_log = []
gc.callbacks.append(lambda phase, info: gc_log.append((phase, info.get("generation"), time.perf_counter())))
for rnd in range(4):
    if GC_OFF:
        gc.collect(); gc.disable()
    gc_log.clear()
    shape = (128, 5, 551, 8)
    arrs = [np.zeros(shape, order="F"), np.zeros(shape, order="F"), np.zeros((128, 551, 8), order="F")]
    t0 = time.perf_counter()
    for a in arrs: dp.pin(a)
    tpin = time.perf_counter() - t0
    ts = []
    for _ in range(25):
        t0 = time.perf_counter(); dp.discrete_adjoint(pcof, False, *arrs); ts.append((time.perf_counter() - t0) * 1e3)
    print(f"round {rnd}: pin {tpin * 1e3:.1f} ms; calls (ms):", " ".join(f"{t:.2f}" for t in ts), "| pages by node", nodes_of(arrs[0]), flush=True)
    full = [(p, g) for p, g, _ in gc_log if p == "start" and g == 2]
    if gc_log:
        spans = []
        st = None
        for p, g, t in gc_log:
            if p == "start": st = (g, t)
            elif st: spans.append((st[0], (t - st[1]) * 1e3)); st = None
        print("   gc runs (generation, ms):", [(g, round(ms, 2)) for g, ms in spans if ms > 0.2 or g == 2], "full collections:", len(full), flush=True)
    gc.enable()
    del arrs
