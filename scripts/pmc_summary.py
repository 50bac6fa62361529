#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs into profiles/<name>.json:
mean counter value per launch of every kernel.  Usage:
    pmc_summary.py out.json FETCH_SIZE=dir_or_csv WRITE_SIZE=dir_or_csv
FETCH_SIZE / WRITE_SIZE are in KB (MI355X_MICROARCH.md 'HBM' gives the gfx950 correction that
bench.py applies: a wide coalesced read is counted at half its size)."""
import csv, glob, json, os, re, sys

def short(name):
    return re.sub(r"\(.*$", "", name).strip()

def load(path, counter):
    files = [path] if path.endswith(".csv") else glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)
    acc = {}
    for f in files:
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != counter:
                continue
            k = short(row["Kernel_Name"])
            v = float(row["Counter_Value"])
            s = acc.setdefault(k, [0.0, 0])
            s[0] += v; s[1] += 1
    return {k: round(s[0] / s[1], 1) for k, s in sorted(acc.items())}

out = {}
for spec in sys.argv[2:]:
    counter, path = spec.split("=", 1)
    out[f"{counter}_KB_mean_per_launch"] = load(path, counter)
json.dump(out, open(sys.argv[1], "w"), indent=1)
print(json.dumps(out, indent=1))
