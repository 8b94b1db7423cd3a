#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE (build container only).

    python tools/oracle/make_golden.py [--only g0,g1,...]

Every array written is an input or an output of the reference's own functions
(`/root/reference/src/*.py`, imported through tools/oracle/refshim.py). No
reference source text is stored. Files are plain .npz (allow_pickle=False).

G0  mathutils known answers           (src/mathutils.py:13-51)
G1  per-point projection + Jacobian blocks, both models, incl. the zero-parameter
    case of tests/test_jacobian.py:19-24   (src/jacobian.py:37-46,147-186)
G2  config 1 (10 views x 54 pts, 9x6 board) radtan + fisheye: detections, poses,
    DLT start point, dense J / JtJ / Jtr / delta at lambda=1e-3, the LM trace
    and the final (sse, A, W, k)       (src/calibrate.py:117-171)
G3  the 15-view 25x18 unit-test dataset (ragged views) of tests/test_calibrate.py:41-45
G4  tests/itest_main.py:12-29 realistic radtan calibration (final A, k)
G5  200 ragged views x <=54 pts: P0, dense delta, Schur inputs (g5k: the same at 1000 views, config 2's scale)
G6  synthetic-generator poses for the bench boards (src/dataset.py:59-95)
G7  DLT homographies and their LM polish (src/linearcalibrate.py:7-58, src/calibrate.py:60-115)
G8  HomographyJacobian.compute and the mathutils helpers exp/skew/unskew/stack/unstack/project/projectStandard
G9  tests/itest_main.py:31-52 noisy radtan calibration: detections (noise model), start point, LM trace, final A, k
G10 the reference's one stored, non-synthetic detection set (tests/test_linearcalibrate.py:266-386 getExampleData: 57
    corners of a real image, with missing corners) through estimateHomography and the LM polish, and the known-answer
    case of tests/test_linearcalibrate.py:55-70 (Hexpected, the reference asserts atol 1e-3)
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np  # noqa: E402

import refshim as ref  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests", "golden")
OUT = os.path.abspath(OUT)


def packDetections(allDetections):
    offs = np.zeros(len(allDetections) + 1, dtype=np.int64)
    for i, (s, m) in enumerate(allDetections):
        offs[i + 1] = offs[i] + s.shape[0]
    sensor = np.vstack([s for s, m in allDetections]).astype(np.float64)
    model = np.vstack([m for s, m in allDetections]).astype(np.float64)
    return offs, sensor, model


def save(name, **arrays):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **arrays)
    print(f"wrote {path}  ({os.path.getsize(path)/1024:.1f} KiB)")


def runTracedRefine(cal, A0, W0, k0, allDetections, maxIters):
    """Runs the reference's own loop; records what it prints per iteration."""
    rows, Pts = [], []

    def rec(self, it, ts, Pt, error, lam):
        rows.append((it, error, lam))
        Pts.append(np.array(Pt, dtype=np.float64).ravel().copy())

    saved = ref.calibrate.Calibrator._printIterationStats
    ref.calibrate.Calibrator._printIterationStats = rec
    try:
        sse, A, W, k = cal.refineCalibrationParameters(
            A0, W0, k0, allDetections, maxIters, shouldPrint=True)
    finally:
        ref.calibrate.Calibrator._printIterationStats = saved
    return sse, A, W, k, np.array(rows, dtype=np.float64), np.array(Pts)


def g0():
    rng = np.random.default_rng(7)
    angles = np.vstack([
        rng.uniform(-180, 180, (40, 3)),
        [[0, 0, 0], [180, 0, 0], [0, 90, 0], [0, -90, 0], [10, 90, 30], [10, -90, 30],
         [1e-7, 0, 0], [0, 1e-9, 2e-7], [-30, 45, 170], [179.999, -89.9, -179.999]],
    ])
    Rs = np.array([ref.mathutils.eulerToRotationMatrix(tuple(a)) for a in angles])
    back = np.array([ref.mathutils.rotationMatrixToEuler(R) for R in Rs])
    save("g0_mathutils.npz", angles=angles, R=Rs, eulerBack=back)


def _blocks(cal, intr, ext, modelPoints):
    jac = cal._jac
    JI = jac._createIntrinsicsJacobianBlock(list(intr), list(ext), modelPoints)
    JE = jac._createExtrinsicsJacobianBlock(list(intr), list(ext), modelPoints)
    P = np.array(list(intr) + list(ext), dtype=np.float64)
    y = cal.projectAllPoints(P, [modelPoints])
    return JI, JE, y


def g1():
    board = ref.checkerboard.Checkerboard(9, 6, 0.05).getCornerPositions()
    rng = np.random.default_rng(11)
    out = {}
    for name, intr in (
            ("radtan", [400, 410, 0.3, 320, 240, -0.5, 0.2, 0.07, -0.03, 0.05]),
            ("fisheye", [803.1, 799.2, -0.2, 700.5, 529.2, -0.155, -0.02, 0.01, -0.03])):
        cal = ref.getCalibrator(name)
        exts, JIs, JEs, ys, pts = [], [], [], [], []
        for j in range(4):
            ext = [rng.uniform(150, 210), rng.uniform(-25, 25), rng.uniform(-180, 180),
                   rng.uniform(-0.25, 0.05), rng.uniform(-0.2, 0.05), rng.uniform(0.5, 1.0)]
            mp = board.copy()
            if j == 3:   # general 3D points (Z != 0)
                mp = mp + rng.uniform(-0.02, 0.02, mp.shape)
            JI, JE, y = _blocks(cal, intr, ext, mp)
            exts.append(ext); JIs.append(JI); JEs.append(JE); ys.append(y); pts.append(mp)
        out[f"{name}_intr"] = np.array(intr, dtype=np.float64)
        out[f"{name}_ext"] = np.array(exts)
        out[f"{name}_modelPoints"] = np.array(pts)
        out[f"{name}_JI"] = np.array(JIs)
        out[f"{name}_JE"] = np.array(JEs)
        out[f"{name}_y"] = np.array(ys)
    # tests/test_jacobian.py:19-24 zero-parameter case
    cal = ref.getCalibrator("radtan")
    intr = [400, 400, 0, 320, 240, -0.5, 0.2, 0, 0, 0]
    ext = [180, 0, 0, 0.1, 0.2, 1.0]
    mp = np.array([[0.1, 0.1, 0], [0.1, 0.2, 0]], dtype=np.float64)
    JI, JE, y = _blocks(cal, intr, ext, mp)
    out.update(zero_intr=np.array(intr, dtype=np.float64), zero_ext=np.array(ext, dtype=np.float64),
               zero_modelPoints=mp, zero_JI=JI, zero_JE=JE, zero_y=y)
    save("g1_blocks.npz", **out)


def _fullProblem(cal, ds, maxIters, tag, storeDenseJ):
    allDetections = ds.getCornerDetectionsInSensorCoordinates()
    Wtrue = np.array(ds.getAllBoardPosesInCamera())
    offs, sensor, model = packDetections(allDetections)
    A0, W0, k0 = cal.estimateCalibrationParameters(allDetections)
    P0 = cal._composeParameterVector(A0, W0, k0).ravel().astype(np.float64)
    allModelPoints = [m for s, m in allDetections]
    t0 = time.time()
    J = cal._jac.compute(P0, allModelPoints)
    y0 = cal.projectAllPoints(P0, allModelPoints)
    r0 = (sensor.reshape(-1, 1) - y0.reshape(-1, 1))
    JTJ = J.T @ J
    JTr = (J.T @ r0).ravel()
    lam = 1e-3
    delta = (np.linalg.inv(JTJ + lam * np.diag(np.diagonal(JTJ))) @ J.T @ r0).ravel()
    err0 = cal._computeReprojectionError(P0, allDetections)
    err1 = cal._computeReprojectionError(P0 + delta, allDetections)
    print(f"[{tag}] one dense step {time.time()-t0:.1f}s err0={err0:.6e} err1={err1:.6e}")
    t0 = time.time()
    sse, A, W, k, rows, Pts = runTracedRefine(cal, A0, W0, k0, allDetections, maxIters)
    print(f"[{tag}] refine {len(rows)} iters {time.time()-t0:.1f}s sse={sse:.3e}")
    Pfinal = cal._composeParameterVector(A, W, k).ravel()
    arrays = dict(
        viewOffsets=offs, sensorPoints=sensor, modelPoints=model, Wtrue=Wtrue,
        Atrue=np.array(ds.getIntrinsicMatrix(), dtype=np.float64),
        ktrue=np.array(ds.getDistortionVector(), dtype=np.float64),
        A0=A0, W0=np.array(W0), k0=np.array(k0, dtype=np.float64), P0=P0,
        y0=y0, JTJ=JTJ, JTr=JTr, delta=delta, err0=np.float64(err0), err1=np.float64(err1),
        maxIters=np.int64(maxIters), traceIterErrLam=rows, tracePt=Pts,
        sseFinal=np.float64(sse), Afinal=A, Wfinal=np.array(W),
        kfinal=np.array(k, dtype=np.float64), Pfinal=Pfinal)
    if storeDenseJ:
        arrays["J"] = J
    return arrays


def g2():
    A = np.array([[400, 0, 320], [0, 400, 240], [0, 0, 1]], dtype=np.float64)
    for name, k, maxIters in (("radtan", (-0.5, 0.2, 0.07, -0.03, 0.05), 100),
                              ("fisheye", (-0.155, -0.02, 0.0, -0.03), 100)):
        cal = ref.getCalibrator(name)
        model = {"radtan": ref.distortion.RadialTangentialModel,
                 "fisheye": ref.distortion.FisheyeModel}[name]()
        cam = ref.virtualcamera.VirtualCamera(A, k, model, 640, 480, None)
        ds = ref.dataset.Dataset(ref.checkerboard.Checkerboard(9, 6, 0.05), cam, 10)
        save(f"g2_config1_{name}.npz", **_fullProblem(cal, ds, maxIters, f"g2-{name}", True))


def g3():
    A = np.array([[400, 0, 320], [0, 400, 240], [0, 0, 1]], dtype=np.float64)
    k = (-0.5, 0.2, 0.07, -0.03, 0.05)
    ds = ref.dataset.createSyntheticDatasetRadTan(A, 640, 480, k, None)
    cal = ref.getCalibrator("radtan")
    arrays = _fullProblem(cal, ds, 100, "g3", False)
    # tests/test_calibrate.py:63-78 round trip and :123-133 zero error at truth
    Wtrue = ds.getAllBoardPosesInCamera()
    Ptrue = cal._composeParameterVector(A, Wtrue, k)
    A2, W2, k2 = cal._decomposeParameterVector(Ptrue)
    arrays.update(Ptrue=Ptrue.ravel(), WfromPtrue=np.array(W2),
                  errAtTruth=np.float64(cal._computeReprojectionError(
                      Ptrue, ds.getCornerDetectionsInSensorCoordinates())))
    save("g3_unittest15.npz", **arrays)


def g4():
    ds = ref.dataset.createRealisticRadTanDataset()
    cal = ref.getCalibrator("radtan")
    save("g4_realistic.npz", **_fullProblem(cal, ds, 100, "g4", False))


def g9():
    """tests/itest_main.py:31-52: the reference's NOISY radial-tangential calibration (sigma = 0.1 px; the noise is
    drawn by src/noise.py:16 from the global generator src/dataset.py:64 re-seeds per view, after the pose draws)."""
    A = np.array([[803.1, 0, 700.5], [0, 803.1, 529.2], [0, 0, 1]], dtype=np.float64)
    k = (-0.25, 0.2, 0.07, -0.03, 0.05)
    ds = ref.dataset.createSyntheticDatasetRadTan(A, 1440, 1080, k, ref.noise.NoiseModel(0.1))
    cal = ref.getCalibrator("radtan")
    arrays = _fullProblem(cal, ds, 100, "g9", False)
    arrays.update(noiseSigma=np.float64(0.1), imageSize=np.array([1440, 1080], dtype=np.int64))
    save("g9_noisy.npz", **arrays)


def g5(M=200):
    A = np.array([[400, 0, 320], [0, 400, 240], [0, 0, 1]], dtype=np.float64)
    k = (-0.5, 0.2, 0.07, -0.03, 0.05)
    cal = ref.getCalibrator("radtan")
    cam = ref.virtualcamera.VirtualCamera(A, k, ref.distortion.RadialTangentialModel(),
                                          640, 480, None)
    ds = ref.dataset.Dataset(ref.checkerboard.Checkerboard(9, 6, 0.05), cam, M)
    allDetections = ds.getCornerDetectionsInSensorCoordinates()
    keep = [d for d in allDetections if d[0].shape[0] >= 6]
    print(f"[g5] {len(keep)} of {M} views kept; pts/view min {min(d[0].shape[0] for d in keep)}"
          f" max {max(d[0].shape[0] for d in keep)}")
    allDetections = keep
    Wtrue = [W for W, d in zip(ds.getAllBoardPosesInCamera(),
                               ds.getCornerDetectionsInSensorCoordinates()) if d[0].shape[0] >= 6]
    offs, sensor, model = packDetections(allDetections)
    # start point: ground truth with a multiplicative perturbation
    Ptrue = cal._composeParameterVector(A, Wtrue, k).ravel()
    rng = np.random.default_rng(0)
    P0 = Ptrue * (1 + 1e-3 * rng.standard_normal(Ptrue.shape))
    allModelPoints = [m for s, m in allDetections]
    t0 = time.time()
    J = cal._jac.compute(P0, allModelPoints)
    print(f"[g5] dense J {J.shape} in {time.time()-t0:.1f}s")
    y0 = cal.projectAllPoints(P0, allModelPoints)
    r0 = sensor.reshape(-1, 1) - y0.reshape(-1, 1)
    JTJ = J.T @ J
    JTr = (J.T @ r0).ravel()
    lam = 1e-3
    delta = (np.linalg.inv(JTJ + lam * np.diag(np.diagonal(JTJ))) @ J.T @ r0).ravel()
    err0 = cal._computeReprojectionError(P0, allDetections)
    err1 = cal._computeReprojectionError(P0 + delta, allDetections)
    L = 10
    save(f"g5_ragged{M}.npz", viewOffsets=offs, sensorPoints=sensor, modelPoints=model,
         Ptrue=Ptrue, P0=P0, y0=y0, JTr=JTr, delta=delta,
         B=JTJ[:L, :L], diagJTJ=np.diagonal(JTJ).copy(),
         err0=np.float64(err0), err1=np.float64(err1),
         Jview0=J[:2 * (offs[1] - offs[0]), :], lam=np.float64(lam))


def g5k():
    """G5 at config 2's scale: 1000 ragged views (one dense reference step, ~10 min and ~12 GB here)."""
    g5(1000)


def g6():
    """Poses + detections of the first views of each bench board (generator pin)."""
    out = {}
    cfgs = {
        "c2": ("radtan", (9, 6, 0.05), [[400, 0, 320], [0, 400, 240], [0, 0, 1]], 640, 480,
               (-0.5, 0.2, 0.07, -0.03, 0.05)),
        "c3": ("fisheye", (20, 10, 0.03), [[803.1, 0, 700.5], [0, 803.1, 529.2], [0, 0, 1]],
               1440, 1080, (-0.155, -0.02, 0.0, -0.03)),
        "c5": ("radtan", (11, 8, 0.04), [[400, 0, 320], [0, 400, 240], [0, 0, 1]], 640, 480,
               (-0.5, 0.2, 0.07, -0.03, 0.05)),
    }
    for tag, (name, (bw, bh, sp), A, w, h, k) in cfgs.items():
        model = {"radtan": ref.distortion.RadialTangentialModel,
                 "fisheye": ref.distortion.FisheyeModel}[name]()
        A = np.array(A, dtype=np.float64)
        cam = ref.virtualcamera.VirtualCamera(A, k, model, w, h, None)
        board = ref.checkerboard.Checkerboard(bw, bh, sp)
        ds = ref.dataset.Dataset(board, cam, 12)
        W = np.array(ds.getAllBoardPosesInCamera())
        # uncropped projection of every corner (bench mode has no crop)
        corners = board.getCornerPositions()
        yfull = np.array([model.projectWithDistortion(A, ref.mathutils.transform(Wi, corners), k)
                          for Wi in W])
        offs, sensor, modelPts = packDetections(ds.getCornerDetectionsInSensorCoordinates())
        out.update({f"{tag}_W": W, f"{tag}_yfull": yfull, f"{tag}_corners": corners,
                    f"{tag}_cropOffsets": offs, f"{tag}_cropSensor": sensor,
                    f"{tag}_cropModel": modelPts, f"{tag}_A": A,
                    f"{tag}_k": np.array(k, dtype=np.float64),
                    f"{tag}_wh": np.array([w, h], dtype=np.int64)})
    save("g6_generator.npz", **out)


def g7():
    """Homography stage (src/linearcalibrate.py:7-58, src/calibrate.py:60-115) on the 15-view
    unit-test dataset and config 1: DLT homographies and their LM-polished versions."""
    out = {}
    A = np.array([[400, 0, 320], [0, 400, 240], [0, 0, 1]], dtype=np.float64)
    k = (-0.5, 0.2, 0.07, -0.03, 0.05)
    cal = ref.calibrate.Calibrator(ref.distortion.RadialTangentialModel())
    ds15 = ref.dataset.createSyntheticDatasetRadTan(A, 640, 480, k, None)
    cam = ref.virtualcamera.VirtualCamera(A, k, ref.distortion.RadialTangentialModel(), 640, 480, None)
    ds10 = ref.dataset.Dataset(ref.checkerboard.Checkerboard(9, 6, 0.05), cam, 10)
    for tag, ds in (("u15", ds15), ("c1", ds10)):
        dets = ds.getCornerDetectionsInSensorCoordinates()
        offs, sensor, model = packDetections(dets)
        Hs = ref.linearcalibrate.estimateHomographies(dets)
        Hdlt = np.array(Hs).copy()          # _refineHomography updates H in place (Pt = H.ravel() is a view)
        Hsref = cal._refineHomographies(Hs, dets)
        out.update({f"{tag}_viewOffsets": offs, f"{tag}_sensorPoints": sensor, f"{tag}_modelPoints": model,
                    f"{tag}_H": Hdlt, f"{tag}_Href": np.array(Hsref)})
    save("g7_homographies.npz", **out)


def g8():
    """Surface helpers: HomographyJacobian.compute (src/jacobian.py:88-121, incl. the call shape of
    tests/test_jacobian.py:79-89) and the mathutils functions exp / skew / unskew / stack / unstack /
    project / projectStandard (src/mathutils.py:59-117,149-192; call shapes of tests/test_mathutils.py)."""
    out = {}
    hj = ref.jacobian.HomographyJacobian()
    g7 = np.load(os.path.join(OUT, "g7_homographies.npz"))
    board = ref.checkerboard.Checkerboard(9, 6, 0.05).getCornerPositions()
    Hs = [np.eye(3), g7["c1_Href"][0], g7["c1_Href"][3], np.array([[1.1, 0.2, 3.0], [-0.3, 0.9, 1.0], [0.01, -0.02, 1.3]])]
    pts = [np.arange(15, dtype=np.float64).reshape(-1, 3), board, board, board + 0.013]
    for i, (H, mp) in enumerate(zip(Hs, pts)):
        out[f"hj{i}_h"] = H.ravel().astype(np.float64)
        out[f"hj{i}_modelPoints"] = mp
        out[f"hj{i}_J"] = np.asarray(hj.compute(H.ravel(), mp), dtype=np.float64)
    rng = np.random.default_rng(5)
    ws = np.vstack([rng.uniform(-3, 3, (12, 3)), [[0, 0, 0], [1e-9, 0, 0], [0, 0, np.pi], [np.pi / 2, 0, 0]]])
    mu = ref.mathutils
    out["exp_w"] = ws
    out["exp_R"] = np.array([mu.exp(mu.skew(mu.col(w))) for w in ws], dtype=np.float64)
    out["skew"] = np.array([mu.skew(mu.col(w)) for w in ws], dtype=np.float64)
    out["unskew"] = np.array([mu.unskew(mu.skew(mu.col(w))) for w in ws], dtype=np.float64)
    Amat = np.arange(9, dtype=np.float64).reshape(3, 3) + 0.5
    out["stack_A"], out["stack_out"] = Amat, mu.stack(Amat)
    out["unstack_out"] = mu.unstack(mu.stack(Amat))
    A = np.array([[400, 0.5, 320], [0, 410, 240], [0, 0, 1]], dtype=np.float64)
    wMc = mu.poseFromRT(mu.eulerToRotationMatrix((170, 12, -33)), (0.1, -0.2, 0.9))
    wX = rng.uniform(-0.3, 0.3, (20, 3))
    out["project_A"], out["project_wMc"], out["project_wX"] = A, wMc, wX
    out["project_u"] = mu.project(A, wMc, wX)
    Xc = rng.uniform(-1, 1, (20, 3)) + np.array([0, 0, 2.5])
    out["projectStandard_X"], out["projectStandard_x"] = Xc, mu.projectStandard(Xc)
    save("g8_surface.npz", **out)


def g10():
    """src/linearcalibrate.py:24-58 (estimateHomography) + src/calibrate.py:69-111 (_refineHomography) on the data the
    reference's own tests hold: getExampleData() (the coordinates are a fixture: stored as arrays) and the 10-point
    known-answer case. The test module is imported from /root/reference/tests for its two data helpers only."""
    sys.path.insert(0, "/root/reference/tests")
    import test_linearcalibrate as tl
    out = {}
    x, X = tl.getExampleData()
    x, X = np.asarray(x, dtype=np.float64), np.asarray(X, dtype=np.float64)
    H = ref.linearcalibrate.estimateHomography(x, X[:, :2])
    cal = ref.calibrate.Calibrator(ref.distortion.RadialTangentialModel())
    Href = cal._refineHomographies([np.array(H).copy()], [(x, X)])[0]
    out.update(ex_x=x, ex_X=X, ex_H=np.array(H, dtype=np.float64), ex_Href=np.array(Href, dtype=np.float64))
    Xk = tl.generateRandomPointsInFrontOfCamera(10)
    Xk = Xk / ref.mathutils.col(Xk[:, 2])
    Hexpected = np.array([[410, 10, 320], [20, 385, 240], [0, 0, 1]], dtype=np.float64)
    xk = (Hexpected @ Xk.T).T
    xk = (xk / ref.mathutils.col(xk[:, 2]))[:, :2]
    Hk = ref.linearcalibrate.estimateHomography(xk, Xk[:, :2])
    assert np.allclose(Hk, Hexpected, atol=1e-3)            # the reference's own assertion
    out.update(ka_X=Xk, ka_x=xk, ka_Hexpected=Hexpected, ka_H=np.array(Hk, dtype=np.float64))
    save("g10_real_detections.npz", **out)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="g0,g1,g2,g3,g4,g5,g6,g7")
    args = ap.parse_args()
    for g in args.only.split(","):
        t0 = time.time()
        globals()[g]()
        print(f"{g} done in {time.time()-t0:.1f}s")
