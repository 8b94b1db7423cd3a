"""Randomised cross-check of the fused / per-view kernel variants (run on the GPU box): for random shard shapes
(uniform and ragged views of 3..700 points, 1..400 views) every combination of J^T J form (tile / block), views per
wave (1..4) and record-head load form (narrow / wide) must give the same normal equations and LM step -- bitwise
where the arithmetic is the same, to rounding otherwise -- and agree with the C oracle. Uniform shards whose views are
whole 4-point groups of at least one batch also go through the STREAM form of the fused kernel with random launch
widths: the wave cuts fall anywhere in the views, batches straddle view boundaries, cut views have two records.
usage: python tools/fuzz_forms.py [trials] [seed]"""
import itertools
import os
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)

CHILD = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, os.environ["FUZZ_ROOT"])
import camera_calibration_amd as cca
d = np.load(os.environ["FUZZ_IN"])
eng = cca.RefineEngine(str(d["name"]), "f64")
eng.setProblem(d["offs"], d["sensor"], d["pts"])
B, E, V, g = eng.normalEquations(d["P0"])
delta = eng.stepDelta(d["P0"], 1e-3)
eng.close()
np.savez(os.environ["FUZZ_OUT"], B=B, E=E, V=V, g=g, delta=delta)
'''


def main():
    import numpy as np
    from camera_calibration_amd import synthetic
    from oracle import c_oracle, calib_oracle as orc
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    worst = 0.0
    for t in range(trials):
        name = ["radtan", "fisheye"][t % 2]
        model = orc.RADTAN if name == "radtan" else orc.FISHEYE
        M = int(rng.integers(1, 400))
        if rng.random() < 0.6:
            sizes = np.full(M, int(rng.choice([3, 4, 5, 31, 32, 33, 54, 63, 64, 65, 68, 88, 127, 128, 129, 132, 200, 256, 257, 513, 700])))
        else:
            sizes = rng.integers(3, int(rng.choice([20, 100, 300, 700])), M)
        offs = np.concatenate(([0], np.cumsum(sizes))).astype(np.int64)
        MN = int(offs[-1])
        cfg = synthetic.CONFIGS["c2" if name == "radtan" else "c3"]
        corners = synthetic.checkerboardCorners(25, 18, 0.02)
        W = synthetic.sampleBoardPosesInCamera(corners, np.arange(7 * t, 7 * t + M))
        Ptrue = synthetic.composeP(cfg["A"], W, cfg["k"])
        pts = np.column_stack((rng.uniform(0, 0.48, MN), rng.uniform(0, 0.34, MN), rng.uniform(-0.01, 0.01, MN)))
        sensor = c_oracle.evaluate(model, Ptrue, offs, None, pts)["y"] + rng.normal(0, 0.05, (MN, 2))
        P0 = Ptrue * (1 + 1e-3 * rng.standard_normal(Ptrue.shape[0]))
        np.savez("/tmp/fuzz_in.npz", name=name, offs=offs, sensor=sensor, pts=pts, P0=P0)
        ref = c_oracle.step(model, P0, offs, sensor, pts, 1e-3)
        outs = {}
        for form, ipw, head in itertools.product(("tile", "block"), ("1", "2", "3", "4"), ("narrow", "wide")):
            if form == "block" and ipw != "1":
                continue                      # several views per wave exist for the tile forms only
            env = dict(os.environ, FUZZ_ROOT=ROOT, FUZZ_IN="/tmp/fuzz_in.npz", FUZZ_OUT="/tmp/fuzz_out.npz",
                       CALIB_GRAM_FORM=form, CALIB_ITEMS_PER_WAVE=ipw, CALIB_HEAD_LOADS=head, CALIB_FUSED_STREAM="0")
            subprocess.run([sys.executable, "-c", CHILD], env=env, check=True)
            outs[(form, ipw, head)] = dict(np.load("/tmp/fuzz_out.npz"))
        n0 = int(sizes[0])
        if np.all(sizes == n0) and n0 % 4 == 0 and n0 >= 64:
            # stream form: 1 .. M waves (a share is at least one view), both head-load forms
            for waves, head in itertools.product(sorted({1, max(1, M // 3), max(1, int(rng.integers(1, M + 1))), M}), ("narrow", "wide")):
                env = dict(os.environ, FUZZ_ROOT=ROOT, FUZZ_IN="/tmp/fuzz_in.npz", FUZZ_OUT="/tmp/fuzz_out.npz",
                           CALIB_FUSED_STREAM="1", CALIB_STREAM_WAVES=str(waves), CALIB_HEAD_LOADS=head)
                subprocess.run([sys.executable, "-c", CHILD], env=env, check=True)
                outs[("stream", str(waves), head)] = dict(np.load("/tmp/fuzz_out.npz"))
        base = outs[("tile", "1", "narrow")]
        for key, o in outs.items():
            for k in ("E", "V"):                                   # per-view blocks
                tol = 0.0 if key[0] == "tile" else 1e-12 * np.abs(base[k]).max()     # block / stream: same sums, another order
                assert np.abs(o[k] - base[k]).max() <= tol, (t, key, k, np.abs(o[k] - base[k]).max())
            assert np.abs(o["B"] - base["B"]).max() <= 1e-12 * np.abs(base["B"]).max(), (t, key, "B")
            err = np.linalg.norm(o["delta"] - ref) / np.linalg.norm(ref)
            worst = max(worst, err)
            assert err < 1e-7, (t, key, "delta vs oracle", err)
        print(f"trial {t}: {name}, {M} views, {MN} points, sizes {sizes.min()}..{sizes.max()}: {len(outs)} variants agree; "
              f"step vs C oracle {max(np.linalg.norm(o['delta'] - ref) / np.linalg.norm(ref) for o in outs.values()):.1e}", flush=True)
    print(f"ok: {trials} trials, worst step error vs the C oracle {worst:.2e}")


if __name__ == "__main__":
    main()
