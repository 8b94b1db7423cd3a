#!/usr/bin/env python3
"""Turn the rocprofv3 CSVs a gpurun call left under gpurun_out/ into the small, tracked
summaries under profiles/.

    python tools/summarize_profile.py --tag r01a_c3 --stats gpurun_out/prof_c3 \
        --fetch gpurun_out/pmc_fetch --write gpurun_out/pmc_write --workload c3

Writes profiles/<tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats, verbatim),
profiles/<tag>_pmc.json (per-kernel FETCH_SIZE / WRITE_SIZE of the LAST dispatch, in bytes,
with the gfx950 corrections of MI355X_MICROARCH.md section HBM: counters are KiB; FETCH_SIZE
counts 64 B per 128-B request of a wide coalesced read, so it is doubled) and updates
profiles/pmc_traffic.json, which bench.py reads for roofline.traffic.
"""
import argparse
import csv
import glob
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def shortName(n):
    n = n.replace("void ", "").replace("calib::", "")
    return n.split("(")[0]


def newest(pattern):
    files = glob.glob(pattern, recursive=True)
    return max(files, key=os.path.getmtime) if files else None


def lastPerKernel(folder, counter):
    # gpurun merges every run's files into the same local folder: only the newest run counts
    files = [newest(os.path.join(folder, "**", "*counter_collection.csv"))]
    out = {}
    for f in files:
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter:
                continue
            out.setdefault(shortName(row["Kernel_Name"]), []).append(float(row["Counter_Value"]))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", required=True)
    ap.add_argument("--stats")
    ap.add_argument("--fetch")
    ap.add_argument("--write")
    ap.add_argument("--workload", default="c3")
    a = ap.parse_args()
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    if a.stats:
        f = newest(os.path.join(a.stats, "**", "*kernel_stats.csv"))
        shutil.copy(f, os.path.join(ROOT, "profiles", f"{a.tag}_kernel_stats.csv"))
    if a.fetch and a.write:
        fetch = lastPerKernel(a.fetch, "FETCH_SIZE")
        write = lastPerKernel(a.write, "WRITE_SIZE")
        summary = {}
        for k in sorted(set(fetch) | set(write)):
            if k.startswith("__amd"):
                continue
            fb = max(fetch.get(k, [0.0])) * 1024 * 2      # KiB, x2 (gfx950 wide-read undercount)
            wb = max(write.get(k, [0.0])) * 1024
            summary[k] = {"fetch_bytes_corrected": fb, "write_bytes": wb, "hbm_bytes_per_launch": fb + wb,
                          "fetch_size_raw_kib_max": max(fetch.get(k, [0.0])),
                          "write_size_raw_kib_max": max(write.get(k, [0.0])),
                          "dispatches_seen": len(fetch.get(k, []))}
        json.dump(summary, open(os.path.join(ROOT, "profiles", f"{a.tag}_pmc.json"), "w"), indent=1)
        tf = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        traffic = json.load(open(tf)) if os.path.exists(tf) else {}
        jac = [v for k, v in summary.items() if k.startswith("jacobian_kernel")]
        gram = [v for k, v in summary.items() if k.startswith("gram_kernel")]
        fused = [v for k, v in summary.items() if k.startswith("fused_kernel")]
        traffic[a.workload] = {
            "source": f"profiles/{a.tag}_pmc.json",
            "jacobian_bytes_per_launch": max(v["hbm_bytes_per_launch"] for v in jac) if jac else None,
            "gram_bytes_per_launch": max(v["hbm_bytes_per_launch"] for v in gram) if gram else None,
            "fused_bytes_per_launch": max(v["hbm_bytes_per_launch"] for v in fused) if fused else None,
        }
        json.dump(traffic, open(tf, "w"), indent=1)
        print(json.dumps(traffic[a.workload]))


if __name__ == "__main__":
    main()
