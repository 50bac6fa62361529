"""Does a large problem evaluated earlier IN THE SAME PROCESS slow the pinned downloads of the reference-shaped cnot3 call?
(bench.py with C5 in front of `with_history`: 3.3-3.9 ms instead of 0.87.)  Prints ms per evaluation with the three arrays
(a) in a fresh process, (b) after a C5 evaluation whose handle was closed, (the plain 1-D copy variant it once compared, QGD_COPY_BLIT, is gone)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from __graft_entry__ import import_package
import bench
qgd = import_package()


def with_history(label):
    prob, ctrl, pcof, target = bench.workload(qgd, 550, 550.0)
    dp = qgd.DeviceProblem(prob, 8); dp.set_controls(ctrl); dp.set_target(target); dp.set_timing(0)
    hist = dp.pin(np.zeros((128, 5, 551, 8), order="F")); lam = dp.pin(np.zeros((128, 5, 551, 8), order="F"))
    forc = dp.pin(np.zeros((128, 551, 8), order="F"))
    for _ in range(3): dp.discrete_adjoint(pcof, False, hist, lam, forc)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): dp.discrete_adjoint(pcof, False, hist, lam, forc)
    torch.cuda.synchronize(); el = (time.perf_counter() - t0) / 20
    print(f"{label}: {el * 1e3:.3f} ms per evaluation with the three arrays; free device memory {torch.cuda.mem_get_info()[0] / 2**30:.1f} GiB", flush=True)
    dp.close()


which = sys.argv[1] if len(sys.argv) > 1 else "c5"
with_history("fresh process")
if which == "c5":
    r = bench.large_n_case(qgd, np, steps=1)
    print("C5 evaluated:", round(r["ms_per_evaluation"], 2), "ms", flush=True)
elif which == "cnot2":
    bench.cnot2_case_gpu(qgd, np, steps=5)
elif which == "alloc":      # only the allocation pattern: 20 GB allocated and freed
    x = torch.empty(20 * 2**30, dtype=torch.uint8, device="cuda"); x.fill_(1); torch.cuda.synchronize(); del x; torch.cuda.empty_cache()
with_history("after " + which)
with_history("after " + which + ", second handle")
