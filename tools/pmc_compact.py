#!/usr/bin/env python3
"""On the GPU box, after tools/pmc_pass.sh <tag>: the mean per dispatch of every counter per kernel ->
gpurun_out/prof/pmc_<tag>.json, and the raw counter_collection.csv files (tens of MiB per pass) are deleted so that
gpurun_out/ stays under what gpurun copies back.   python tools/pmc_compact.py <tag>"""
import csv, glob, json, os, shutil, sys

ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(n):
    return n.replace("void ", "").replace("calib::", "").split("(")[0].split("<")[0]


def collect(tag):
    acc = {}
    for f in sorted(glob.glob(os.path.join(ROOT, f"gpurun_out/pmc_{tag}_*/**/*counter_collection.csv"), recursive=True)):
        per = {}
        for r in csv.DictReader(open(f)):
            key = (short(r["Kernel_Name"]), r["Counter_Name"], r["Dispatch_Id"])
            per[key] = per.get(key, 0.0) + float(r["Counter_Value"])
        for (k, c, _), v in per.items():
            acc.setdefault(k, {}).setdefault(c, []).append(v)
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items() if not k.startswith("__amd")}


if __name__ == "__main__":
    tag = sys.argv[1]
    out = os.path.join(ROOT, "gpurun_out", "prof", f"pmc_{tag}.json")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    json.dump(collect(tag), open(out, "w"), indent=1, sort_keys=True)
    for d in glob.glob(os.path.join(ROOT, f"gpurun_out/pmc_{tag}_*")):
        if os.path.isdir(d):
            shutil.rmtree(d)
    print(f"wrote {out}")
