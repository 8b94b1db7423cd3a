"""Stress of fused_stream_kernel's boundary handling (run on the GPU box): many (points per view, views, launch width)
combinations -- every wave cut position, batches that straddle at every group, shares of exactly one view, single
waves -- each compared with the one-view-per-wave forms on the same inputs: normal equations to 1e-12, the LM step to
1e-9.   usage: python tools/stress_stream.py [cases] [seed]"""
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np                                    # noqa: E402
import camera_calibration_amd as cca                  # noqa: E402
from camera_calibration_amd import synthetic         # noqa: E402


def run(name, offs, s, m, P0, stream, waves):
    os.environ["CALIB_FUSED_STREAM"] = "1" if stream else "0"
    os.environ["CALIB_STREAM_WAVES"] = str(waves)
    eng = cca.RefineEngine(name, "f64")
    eng.setProblem(offs, s, m)
    assert (eng.fusedForm()[0] > 0) == stream
    out = eng.normalEquations(P0) + (eng.stepDelta(P0, 1e-3),)
    eng.close()
    return out


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    boards = [(8, 8), (17, 4), (9, 8), (11, 8), (10, 10), (16, 8), (11, 12), (20, 10), (16, 16), (32, 16)]   # 64 .. 512 points
    worst = 0.0
    for t in range(cases):
        name = ["radtan", "fisheye"][t % 2]
        w, h = boards[int(rng.integers(len(boards)))]
        views = int(rng.integers(1, 70))
        waves = int(rng.integers(1, views + 1))
        cfg = dict(synthetic.CONFIGS["c2" if name == "radtan" else "c3"], board=(w, h, 0.6 / max(w, h)))
        sh = synthetic.makeShard(cfg, viewStart=int(rng.integers(0, 1000)), numViews=views, noiseSigma=0.05)
        offs, s, m, P0 = sh["viewOffsets"], sh["sensorPoints"], sh["modelPoints"], sh["P0"]
        a = run(name, offs, s, m, P0, False, waves)
        b = run(name, offs, s, m, P0, True, waves)
        for k, (x, y) in enumerate(zip(a[:4], b[:4])):
            err = np.abs(x - y).max() / np.abs(x).max()
            worst = max(worst, err)
            assert err <= 1e-12, (t, name, w * h, views, waves, "BEVg"[k], err)
        err = np.linalg.norm(a[4] - b[4]) / np.linalg.norm(a[4])
        assert err < 1e-9, (t, name, w * h, views, waves, "delta", err)
        if t % 25 == 0:
            print(f"case {t}: {name}, {views} views x {w * h} points, {waves} waves: ok", flush=True)
    print(f"ok: {cases} cases, worst relative difference of the normal equations {worst:.2e}")


if __name__ == "__main__":
    main()
