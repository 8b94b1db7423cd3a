#!/usr/bin/env python3
"""profiles/pmc_fused.json (what bench.py quotes as measured counters, with their source) and
profiles/<round>_<workload>_pmc.json from the counter passes of tools/pmc_pass.sh:

    python tools/make_pmc_fused.py --round r02 --workload c3 --tag c3 --batches 40000 --model fisheye

--batches = 64-lane batches one fused launch evaluates (one view item per wave: views x ceil(points per view / 64);
stream form: the sum over the waves of ceil(points of the wave's share / 64)). Without --batches the count is derived
from the launch shape the profiled build reported for this shard -- `config.fused_form` {share, waves} and the shard's
views x points in gpurun_out/prof/bench_<tag>.json -- and share / waves are recorded, so that bench.py can tell when it
runs with another launch width (another CU count, CALIB_STREAM_WAVES) than the counters were taken at.
Counters are the mean over the dispatches of a kernel; FETCH_SIZE / WRITE_SIZE come in KiB, and on gfx950
FETCH_SIZE counts 64 B per 128-B request of a wide coalesced read, so it is doubled (MI355X_MICROARCH.md, HBM)."""
import argparse, csv, glob, json, os, subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(n):
    return n.replace("void ", "").replace("calib::", "").split("(")[0].split("<")[0]


def collect(tag):
    """mean per dispatch of every counter per kernel: from the compact file tools/pmc_compact.py wrote on the GPU box
    (gpurun_out/prof/pmc_<tag>.json), else from the raw counter_collection.csv files"""
    compact = os.path.join(ROOT, "gpurun_out", "prof", f"pmc_{tag}.json")
    if os.path.exists(compact):
        return json.load(open(compact))
    acc = {}
    for f in sorted(glob.glob(os.path.join(ROOT, f"gpurun_out/pmc_{tag}_*/**/*counter_collection.csv"), recursive=True)):
        per = {}
        for r in csv.DictReader(open(f)):
            key = (short(r["Kernel_Name"]), r["Counter_Name"], r["Dispatch_Id"])
            per[key] = per.get(key, 0.0) + float(r["Counter_Value"])
        for (k, c, _), v in per.items():
            acc.setdefault(k, {}).setdefault(c, []).append(v)
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items() if not k.startswith("__amd")}


def sourcesDigest():
    """sha256 over the kernel sources: bench.py compares it with the build it runs, so that a line quoting counters of
    another build says so (there is no .git on the GPU box to compare commits with)"""
    import hashlib
    h = hashlib.sha256()
    import re
    for f in ("kernels.hpp", "point_model.hpp"):      # the device code, comments and white space aside (as bench.py)
        text = open(os.path.join(ROOT, "camera-calibration_amd", "csrc", f)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        text = re.sub(r"//[^\n]*", "", text)
        h.update(re.sub(r"\s+", "", text).encode())
    return h.hexdigest()[:16]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--round", default="r03")
    ap.add_argument("--workload", required=True)
    ap.add_argument("--tag", required=True)
    ap.add_argument("--batches", type=int, default=0)
    ap.add_argument("--model", required=True)
    a = ap.parse_args()
    k = collect(a.tag)
    form = None
    if a.batches <= 0:
        line = json.loads(open(os.path.join(ROOT, "gpurun_out", "prof", f"bench_{a.tag}.json")).read())
        cfg = line["config"]
        form = cfg["fused_form"]
        views, n = int(cfg["views_per_gpu"]), int(cfg["points_per_view"])
        if form["share"] > 0:
            pts, per = views * n, 4 * form["share"]
            full, rest = divmod(pts, per)
            a.batches = full * -(-per // 64) + (-(-rest // 64) if rest else 0)
        else:
            a.batches = views * -(-n // 64)
    commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
    out = os.path.join(ROOT, "profiles", f"{a.round}_{a.workload}_pmc.json")
    json.dump({"commit": commit, "command": f"tools/pmc_pass.sh {a.tag} (bench.py --steps 10, workload {a.workload})",
               "mean_per_dispatch": k}, open(out, "w"), indent=1, sort_keys=True)
    fused = k.get("fused_stream_kernel") or k.get("fused_kernel", {})
    jac = k.get("jacobian_kernel", {})
    f64 = 64.0 * (2 * fused.get("SQ_INSTS_VALU_FMA_F64", 0) + fused.get("SQ_INSTS_VALU_MUL_F64", 0)
                  + fused.get("SQ_INSTS_VALU_ADD_F64", 0) + fused.get("SQ_INSTS_VALU_TRANS_F64", 0)) / a.batches
    traffic = lambda d: (d.get("FETCH_SIZE", 0) * 1024 * 2 + d.get("WRITE_SIZE", 0) * 1024) if d else None
    pf = os.path.join(ROOT, "profiles", "pmc_fused.json")
    d = json.load(open(pf)) if os.path.exists(pf) else {}
    src = f"profiles/{a.round}_{a.workload}_pmc.json @ {commit}"
    d["kernel_sources_sha256_16"] = sourcesDigest()
    d[a.workload] = {"valu_f64_flops_per_batch": f64 or None, "fused_hbm_bytes_per_launch": traffic(fused),
                     "jacobian_hbm_bytes_per_launch": traffic(jac), "source": src,
                     "valu_instructions_per_batch": (fused.get("SQ_INSTS_VALU", 0) - fused.get("SQ_INSTS_MFMA", 0)) / a.batches,
                     "batches_per_launch": a.batches, "fused_form": form,
                     "mfma_busy_cycles_per_launch": fused.get("SQ_VALU_MFMA_BUSY_CYCLES"),
                     "lds_bank_conflict_frac": (fused.get("SQ_LDS_BANK_CONFLICT", 0) / fused["SQ_LDS_IDX_ACTIVE"])
                     if fused.get("SQ_LDS_IDX_ACTIVE") else None,
                     "wait_inst_any_frac": (fused.get("SQ_WAIT_INST_ANY", 0) / fused["SQ_WAVE_CYCLES"])
                     if fused.get("SQ_WAVE_CYCLES") else None}
    d.setdefault("by_model", {})[a.model] = {"valu_f64_flops_per_batch": f64 or None, "source": src}
    json.dump(d, open(pf, "w"), indent=1, sort_keys=True)
    print(json.dumps(d[a.workload]))


if __name__ == "__main__":
    main()
