#!/usr/bin/env python3
"""Kernel timing experiments: HIP-event time of the jacobian and gram kernels for a workload,
ignoring LM outcome (used with tuning env knobs). Not part of the product or the bench."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import camera_calibration_amd as cca
from camera_calibration_amd import synthetic

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="c3")
ap.add_argument("--views", type=int, default=None)
ap.add_argument("--steps", type=int, default=30)
a = ap.parse_args()
cfg = synthetic.CONFIGS[a.workload]
sh = synthetic.makeShard(cfg, numViews=a.views or cfg["views"], noiseSigma=0.1)
eng = cca.RefineEngine(cfg["model"], cfg["dtype"])
eng.setProblem(sh["viewOffsets"], sh["sensorPoints"], sh["modelPoints"])
eng.lmBegin(sh["P0"], a.steps + 10, lamMin=0.0, lamMax=float("inf"), errMin=-float("inf"))
eng.lmRun(4)
eng.lmDone()
if not os.environ.get("KEXP_NOPROF"):
    eng.profileEnable(True)
import time
t0 = time.perf_counter()
eng.lmRun(a.steps)
eng.lmDone()
el = time.perf_counter() - t0
j, jn = eng.profileRead(0)
g, gn = eng.profileRead(1)
f, fn = eng.profileRead(2)
MN = int(sh["viewOffsets"][-1])
print(f"{a.workload} MN={MN} ms/iter {el/a.steps*1e3:.4f}  jac {j/max(jn,1)*1e3:.1f} us  gram {g/max(gn,1)*1e3:.1f} us  "
      f"fused {f/max(fn,1)*1e3:.1f} us  launches {jn} {gn} {fn}")
eng.close()
