#!/usr/bin/env python3
"""Per-kernel VGPRs / scratch / occupancy / LDS of the library's device code, from hipcc's own remarks
(-Rpass-analysis=kernel-resource-usage).   python tools/resource_usage.py [extra hipcc flags...] [--filter SUBSTR]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    args = sys.argv[1:]
    flt = None
    if "--filter" in args:
        i = args.index("--filter")
        flt = args[i + 1]
        del args[i:i + 2]
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-c", "-Wno-unused-function",
           "-Rpass-analysis=kernel-resource-usage", *args, "-o", "/dev/null", "calib_lm.hip"]
    out = subprocess.run(cmd, cwd=os.path.join(ROOT, "camera-calibration_amd", "csrc"), capture_output=True, text=True).stderr
    rows, cur = [], None
    for ln in out.splitlines():
        m = re.search(r"remark:\s+(.*?) \[-Rpass", ln)
        if not m:
            if "error" in ln:
                print(ln)
            continue
        t = m.group(1).strip()
        if t.startswith("Function Name:"):
            name = subprocess.run(["c++filt", t.split(":", 1)[1].strip()], capture_output=True, text=True).stdout.strip()
            name = re.sub(r"\(.*", "", name.replace("void calib::", "").replace("calib::", ""))
            cur = {"name": name}
            rows.append(cur)
        elif cur is not None and ":" in t:
            k, v = t.split(":", 1)
            cur[k.strip()] = v.strip()
    print(f"{'kernel':70s} {'VGPR':>5s} {'AGPR':>5s} {'scratch':>8s} {'waves/SIMD':>10s} {'LDS':>7s}")
    for r in rows:
        if flt and flt not in r["name"]:
            continue
        print(f"{r['name'][:70]:70s} {r.get('VGPRs', '?'):>5s} {r.get('AGPRs', '?'):>5s} {r.get('ScratchSize [bytes/lane]', '?'):>8s} "
              f"{r.get('Occupancy [waves/SIMD]', '?'):>10s} {r.get('LDS Size [bytes/block]', '?'):>7s}")


if __name__ == "__main__":
    main()
