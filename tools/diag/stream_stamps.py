"""Where a wave of fused_stream_kernel spends its life: per-phase s_memtime deltas from the diagnostic build
(tools/diag/build_stream_stamps.sh), averaged over the waves of one launch, plus when the waves start and end.
usage: python tools/diag/stream_stamps.py [workload] [views]"""
import ctypes
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
os.environ["CALIB_LM_LIBRARY"] = os.path.join(ROOT, "tools", "diag", "lib", "stream_stamps", "libcalib_lm.so")
os.environ.setdefault("CALIB_FUSED_STREAM", "1")
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
NW, NS = 8192, 16
buf = (ctypes.c_ulonglong * (NW * NS))()
lib.calib_debug_stream_stamps(buf, NW)          # clears
eng.lmRun(1)
eng.lmDone()
lib.calib_debug_stream_stamps(buf, NW)
v = np.frombuffer(buf, dtype=np.uint64).reshape(NW, NS).astype(np.float64)
v = v[v[:, 14] == 1]
names = ["prologue", "wait for the batch's points", "Jacobian, one view (scalar constants)", "Jacobian, two views (staged constants)",
         "chunk swaps + slab stores", "contraction, unrolled pass", "contraction, checked pass", "record of a finished view",
         "last park + record", "workgroup barrier", "partial of B", "-"]
tot = v[:, :12].sum()
nw = v.shape[0]
print(f"{wl}: {nw} waves stamped; {tot / nw:.0f} s_memtime ticks per wave")
for n, x in zip(names, v[:, :12].sum(axis=0)):
    print(f"  {n:42s} {x / nw:9.0f} ticks/wave  {100 * x / tot:5.1f} %")
t0, t1 = v[:, 12], v[:, 13]
base = t0.min()
print(f"  wave start: min 0, median {np.median(t0) - base:.0f}, max {t0.max() - base:.0f};  wave end: min {t1.min() - base:.0f}, "
      f"median {np.median(t1) - base:.0f}, max {t1.max() - base:.0f} ticks after the first start")
life = t1 - t0
print(f"  wave lifetime: min {life.min():.0f} median {np.median(life):.0f} max {life.max():.0f}")
