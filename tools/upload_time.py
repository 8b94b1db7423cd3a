#!/usr/bin/env python3
"""calib_set_problem timing (pack + upload of the correspondences), first and repeated calls."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import camera_calibration_amd as cca
from camera_calibration_amd import synthetic
for w, views in (("c3", None), ("c5", 125000)):
    cfg = synthetic.CONFIGS[w]
    sh = synthetic.makeShard(cfg, numViews=views or cfg["views"], noiseSigma=0.1)
    MN = int(sh["viewOffsets"][-1])
    for dtype in ("f64", "f32"):
        eng = cca.RefineEngine(cfg["model"], dtype)
        ts = []
        for rep in range(4):
            t0 = time.perf_counter()
            eng.setProblem(sh["viewOffsets"], sh["sensorPoints"], sh["modelPoints"])
            ts.append(time.perf_counter() - t0)
        ev = eng.evaluate(sh["Ptrue"])["sse"]
        print(f"{w} {dtype} MN={MN} ({MN*40/1e6:.0f} MB): set_problem {[round(t*1e3,2) for t in ts]} ms -> {MN*40/min(ts)/1e9:.1f} GB/s best; sse at truth {ev:.3e}")
        eng.close()
