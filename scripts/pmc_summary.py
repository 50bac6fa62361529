#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs into profiles/<name>.json: mean counter value per launch of
every kernel.
    pmc_summary.py out.json FETCH_SIZE=dir_or_csv WRITE_SIZE=dir_or_csv
    pmc_summary.py out.json --mfma dir          (pass with SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA)
FETCH_SIZE / WRITE_SIZE are in KB, as the counters report them (MI355X_MICROARCH.md 'HBM' gives the gfx950 correction
that bench.py applies: a wide coalesced read is counted at half its size; FETCH and WRITE need separate passes).
mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs): the share of the kernel's cycles in which
a SIMD's MFMA pipe is busy, averaged over the chip (an f64 16x16x4 MFMA holds the pipe for 64 cycles)."""
import csv, glob, json, os, re, sys


def short(name):
    return re.sub(r"\(.*$", "", name).strip()


def rows_of(path):
    files = [path] if path.endswith(".csv") else glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)
    for f in files:
        yield from csv.DictReader(open(f))


def load(path, counter):
    acc = {}
    for row in rows_of(path):
        if row.get("Counter_Name") != counter:
            continue
        s = acc.setdefault(short(row["Kernel_Name"]), [0.0, 0])
        s[0] += float(row["Counter_Value"]); s[1] += 1
    return {k: round(s[0] / s[1], 1) for k, s in sorted(acc.items())}


out = {}
if sys.argv[2] == "--mfma":
    per = {}
    for row in rows_of(sys.argv[3]):
        d = per.setdefault(short(row["Kernel_Name"]).replace("void ", ""), {})
        c = d.setdefault(row["Counter_Name"], [0.0, 0])
        c[0] += float(row["Counter_Value"]); c[1] += 1
    kern = {}
    for k, d in per.items():
        mean = {c: v[0] / v[1] for c, v in d.items()}
        gui = mean.get("GRBM_GUI_ACTIVE", 0.0)
        kern[k] = {"launches": max(v[1] for v in d.values()),
                   "mfma_util": round(mean.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui / 8 * 1024), 3) if gui else None,
                   "SQ_INSTS_VALU": mean.get("SQ_INSTS_VALU"), "SQ_INSTS_MFMA": mean.get("SQ_INSTS_MFMA"),
                   "GRBM_GUI_ACTIVE": round(gui, 1)}
    out = {"notes": "means per launch; mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs)", "kernels": kern}
else:
    for spec in sys.argv[2:]:
        counter, path = spec.split("=", 1)
        out[f"{counter}_KB_mean_per_launch"] = load(path, counter)
json.dump(out, open(sys.argv[1], "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
