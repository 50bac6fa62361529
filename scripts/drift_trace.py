"""Where does the per-evaluation time go over the first 100 evaluations of a process (the `value` vs `settled` question)?
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/drift -- python3 scripts/drift_trace.py run
    python3 scripts/drift_trace.py show gpurun_out/drift
prints, per evaluation: period (start to next start), device-busy time (sum of kernel durations), gap to the next evaluation,
and the durations of the three largest kernels -- kernel times that shrink point at the card's clocks, gaps at the host."""
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
    for _ in range(100):
        dp.discrete_adjoint(pcof)
    torch.cuda.synchronize()
else:
    f = glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    starts = [i for i, r in enumerate(rows) if "k_tables" in r["Kernel_Name"]]
    print("eval  period  busy   gap   inverse  build  gradpoint")
    for e, (a, b) in enumerate(zip(starts[:-1], starts[1:])):
        ev = rows[a:b]
        dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        t0, tend, tnext = int(ev[0]["Start_Timestamp"]), int(ev[-1]["End_Timestamp"]), int(rows[b]["Start_Timestamp"])
        pick = lambda name: sum(dur(r) for r in ev if name in r["Kernel_Name"])
        if e < 30 or e % 10 == 0:
            print(f"{e:4d} {(tnext - t0) / 1e3:7.1f} {sum(dur(r) for r in ev):6.1f} {(tnext - tend) / 1e3:5.1f}   {pick('k_inverse'):6.1f} {pick('k_build_LR'):6.1f} {pick('k_gradpoint'):6.1f}")
