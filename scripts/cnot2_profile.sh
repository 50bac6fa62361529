#!/bin/bash
# rocprofv3 kernel stats of scripts/cnot2_timing.py (order 8 only: 3 + 5 + 200 = 208 evaluations)
#   -> gpurun_out/cnot2prof/{kernel_stats.csv, cnot2_launches.json}
# cnot2_launches.json (copied to profiles/<round>_cnot2_launches.json) is where bench.py's cnot2.roofline reads the number
# of dependent kernel launches of one evaluation from.
set -u
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/cnot2prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export QGD_CNOT2_ORDER8=1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $REPO/scripts/cnot2_timing.py > $OUT/out.txt 2> $OUT/err.txt
cp $(ls $OUT/stats/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
rm -rf $OUT/stats
python3 - "$OUT/kernel_stats.csv" "$OUT/cnot2_launches.json" <<'PY'
import csv, json, sys
EVALS = 208
rows = list(csv.DictReader(open(sys.argv[1])))
ours = {r["Name"]: int(r["Calls"]) for r in rows if "k_" in r["Name"] and "rocclr" not in r["Name"]}
other = {r["Name"]: int(r["Calls"]) for r in rows if r["Name"] not in ours}
per = {k.split("(")[0].replace("void ", ""): round(v / EVALS, 3) for k, v in ours.items()}
out = {"command": "QGD_CNOT2_ORDER8=1 rocprofv3 --kernel-trace --stats -- python3 scripts/cnot2_timing.py", "evaluations": EVALS,
       "launches_per_evaluation": round(sum(ours.values()) / EVALS, 2), "kernels_per_evaluation": per,
       "runtime_copy_kernels_per_evaluation": round(sum(other.values()) / EVALS, 2),
       "device_us_per_evaluation": round(sum(int(r["TotalDurationNs"]) for r in rows) / EVALS / 1e3, 1)}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out))
PY
cut -d, -f1-4 $OUT/kernel_stats.csv | cut -c1-150 | head -24; tail -3 $OUT/out.txt
