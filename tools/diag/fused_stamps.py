"""Where a wave of the fused kernel spends its life: per-phase s_memtime deltas from the diagnostic build
(tools/diag/build_diag.sh -> tools/diag/lib/stamps), averaged over the waves of one launch.
usage: python tools/diag/fused_stamps.py [workload] [views]"""
import ctypes
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
os.environ["CALIB_LM_LIBRARY"] = os.path.join(ROOT, "tools", "diag", "lib", "stamps", "libcalib_lm.so")
import numpy as np                                   # noqa: E402
import camera_calibration_amd as cca                 # noqa: E402
from camera_calibration_amd import synthetic, _native  # noqa: E402

lib = _native.loadLibrary()
wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
views = int(sys.argv[2]) if len(sys.argv) > 2 else None
cfg = synthetic.CONFIGS[wl]
sh = synthetic.makeShard(cfg, numViews=views or cfg["views"], noiseSigma=0.1)
eng = cca.RefineEngine(cfg["model"], cfg["dtype"])
eng.setProblem(sh["viewOffsets"], sh["sensorPoints"], sh["modelPoints"])
eng.lmBegin(sh["P0"], 100, lamMin=0.0, lamMax=float("inf"), errMin=-float("inf"))
eng.lmRun(10)
eng.lmDone()
buf = (ctypes.c_double * 10)()
lib.calib_debug_stamps(buf)          # clears
eng.lmRun(1)
eng.lmDone()
lib.calib_debug_stamps(buf)
v = np.array(list(buf), dtype=np.float64)
waves = v[9]
names = ["prologue", "wait for the batch's points", "Jacobian (VALU)", "slab stores", "slab loads + MFMA", "park tiles",
         "workgroup barrier", "partial of B", "record"]
tot = v[:9].sum()
print(f"{wl}: {waves:.0f} waves stamped; {tot / waves:.0f} s_memtime cycles per wave")
for n, x in zip(names, v[:9]):
    print(f"  {n:28s} {x / waves:9.0f} cycles/wave  {100 * x / tot:5.1f} %")
