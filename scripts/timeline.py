"""Kernel timeline of ONE cnot3 evaluation from a rocprofv3 kernel trace:
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python3 scripts/timeline.py run
    python3 scripts/timeline.py show gpurun_out/tl
prints, for the last evaluation of the run, every kernel with its start / end (us, relative to the first) and queue."""
import csv, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if sys.argv[1] == "run":
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np, torch
    from __graft_entry__ import import_package
    import bench
    qgd = import_package()
    prob, ctrl, pcof, target = bench.workload(qgd, 550, 550.0)
    dp = qgd.DeviceProblem(prob, 8); dp.set_controls(ctrl); dp.set_target(target); dp.set_timing(0)
    if len(sys.argv) > 2 and sys.argv[2] == "history":      # the reference-shaped call with the three output arrays, pinned
        hist = dp.pin(np.zeros((128, 5, 551, 8), order="F")); lam = dp.pin(np.zeros((128, 5, 551, 8), order="F"))
        forc = dp.pin(np.zeros((128, 551, 8), order="F"))
        for _ in range(6):
            dp.discrete_adjoint(pcof, False, hist, lam, forc)
    else:
        for _ in range(6):
            dp.discrete_adjoint(pcof)
    torch.cuda.synchronize()
else:
    f = glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    for fm in glob.glob(os.path.join(sys.argv[2], "**", "*memory_copy_trace.csv"), recursive=True):
        for r in csv.DictReader(open(fm)):
            rows.append({"Kernel_Name": "COPY " + r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", "?")) + " B", "Start_Timestamp": r["Start_Timestamp"],
                         "End_Timestamp": r["End_Timestamp"], "Queue_Id": "-", "Grid_Size": "-"})
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # last evaluation = from the last k_tables* kernel on
    idx = max(i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("void k_tables") or r["Kernel_Name"].startswith("k_tables"))
    ev = rows[idx:]
    t0 = int(ev[0]["Start_Timestamp"])
    for r in ev:
        s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
        print(f"{s:8.1f} {e:8.1f} {e - s:7.1f}  q{r.get('Queue_Id', '?'):>3}  grid {r.get('Grid_Size', '?'):>7}  {r['Kernel_Name'][:60]}")
