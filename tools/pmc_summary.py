#!/usr/bin/env python3
"""Mean per-dispatch value of every counter for kernels whose name contains a substring, from the
counter_collection.csv files under gpurun_out/pmc_<tag>_*: python tools/pmc_summary.py <tag> [substring]"""
import csv, glob, sys
tag = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "fused"
acc = {}
for f in sorted(glob.glob(f"gpurun_out/pmc_{tag}_*/**/*counter_collection.csv", recursive=True)):
    per = {}
    for r in csv.DictReader(open(f)):
        if pat not in r["Kernel_Name"]:
            continue
        per.setdefault((r["Counter_Name"], r["Dispatch_Id"]), 0.0)
        per[(r["Counter_Name"], r["Dispatch_Id"])] += float(r["Counter_Value"])
    for (name, _), v in per.items():
        acc.setdefault(name, []).append(v)
for name in sorted(acc):
    v = acc[name]
    print(f"{name:32s} n={len(v):4d} mean={sum(v)/len(v):16.1f}")
