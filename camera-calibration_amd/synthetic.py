"""Synthetic checkerboard datasets of the benchmark shapes (BASELINE.json configs).

Restates the reference's generator -- src/dataset.py:59-95 (per-view seeded pose
sampling), src/checkerboard.py:9-17 (corner grid), src/virtualcamera.py:42-55
(projection through the distortion model) -- with two differences that the
benchmark configs ask for: no image crop (every view keeps all N corners) and
views addressed by a global index so that each GPU shard generates its own
range. The projection of the corners runs on the device (RefineEngine), like
every other evaluation of the camera model in this package.
"""
import numpy as np

from . import engine
from . import mathutils as mu

RADTAN_A = np.array([[400.0, 0, 320], [0, 400, 240], [0, 0, 1]])
RADTAN_K = (-0.5, 0.2, 0.07, -0.03, 0.05)               # tests/test_calibrate.py:35-42
FISHEYE_A = np.array([[803.1, 0, 700.5], [0, 803.1, 529.2], [0, 0, 1]])
FISHEYE_K = (-0.155, -0.02, 0.0, -0.03)                 # tests/itest_main.py:55-61

# name -> model, dtype, board (w, h, spacing), views, A, k      (BASELINE.json "configs")
CONFIGS = {
    "c1": dict(model="radtan", dtype="f64", board=(9, 6, 0.05), views=10, A=RADTAN_A, k=RADTAN_K),
    "c2": dict(model="radtan", dtype="f64", board=(9, 6, 0.05), views=1000, A=RADTAN_A, k=RADTAN_K),
    "c3": dict(model="fisheye", dtype="f64", board=(20, 10, 0.03), views=10000, A=FISHEYE_A, k=FISHEYE_K),
    "c4": dict(model="radtan", dtype="f32", board=(9, 6, 0.05), views=100000, A=RADTAN_A, k=RADTAN_K),
    "c5": dict(model="radtan", dtype="f64", board=(11, 8, 0.04), views=1000000, A=RADTAN_A, k=RADTAN_K),
}

_minDistanceFromBoard, _maxDistanceFromBoard = 0.5, 1.0   # src/dataset.py:18-21
_rollPitchBounds, _yawBounds = (-30, +30), (-180, +180)


def checkerboardCorners(numCornersWidth, numCornersHeight, spacing):
    """(W*H, 3) corners on z = 0, x fastest (src/checkerboard.py:9-17)"""
    i, j = np.meshgrid(np.arange(numCornersWidth), np.arange(numCornersHeight))
    return np.stack((i.ravel() * spacing, j.ravel() * spacing,
                     np.zeros(numCornersWidth * numCornersHeight)), axis=1).astype(np.float64)


def sampleBoardPosesInCamera(corners, viewIndices):
    """(len(viewIndices), 4, 4) board-in-camera poses.

    View i draws, from the legacy generator seeded with i (src/dataset.py:64-70): the corner
    to aim at, roll, pitch, yaw and the distance; the camera pose is
    T(corner) R(180,0,0) . R(roll,pitch,yaw) . T(0,0,-d) (src/dataset.py:84-95) and the board
    pose its inverse (:76)."""
    viewIndices = np.asarray(viewIndices, dtype=np.int64)
    n = viewIndices.shape[0]
    aim = np.empty(n, dtype=np.int64)
    ang = np.empty((n, 3))
    dist = np.empty(n)
    for j, vi in enumerate(viewIndices):
        rs = np.random.RandomState(int(vi))
        aim[j] = rs.choice(corners.shape[0])
        ang[j, 0] = rs.uniform(*_rollPitchBounds)
        ang[j, 1] = rs.uniform(*_rollPitchBounds)
        ang[j, 2] = rs.uniform(*_yawBounds)
        dist[j] = rs.uniform(_minDistanceFromBoard, _maxDistanceFromBoard)
    Rflip = mu.eulerToRotationMatrix((180.0, 0.0, 0.0))
    R = Rflip @ mu.eulerToRotationMatrices(ang)           # board_R_camera
    t = corners[aim] - dist[:, None] * R[:, :, 2]         # camera position in the board frame
    Rinv = np.transpose(R, (0, 2, 1))
    return mu.posesFromRT(Rinv, -np.einsum("nij,nj->ni", Rinv, t))


def composeP(A, W, k):
    """parameter vector (K,) from A, poses (M,4,4), k  (src/calibrate.py:199-229)"""
    W = np.asarray(W, dtype=np.float64).reshape(-1, 4, 4)
    shared = np.array([A[0, 0], A[1, 1], A[0, 1], A[0, 2], A[1, 2]] + list(k), dtype=np.float64)
    ext = np.hstack((mu.rotationMatricesToEuler(W[:, :3, :3]), W[:, :3, 3]))
    return np.concatenate((shared, ext.ravel()))


def makeShard(config, viewStart=0, numViews=None, noiseSigma=0.0, noiseSeed=12345, device=0,
              perturb=1e-3, perturbSeed=0):
    """Generate views [viewStart, viewStart+numViews) of a benchmark config.

    -> dict(viewOffsets, sensorPoints, modelPoints, Ptrue, P0, model, dtype)
    P0 = Ptrue * (1 + perturb * N(0,1)) with the shared parameters perturbed identically on every
    shard (they are one set of unknowns) and the extrinsics per global view index."""
    cfg = CONFIGS[config] if isinstance(config, str) else config
    numViews = cfg["views"] if numViews is None else int(numViews)
    corners = checkerboardCorners(*cfg["board"])
    N = corners.shape[0]
    W = sampleBoardPosesInCamera(corners, np.arange(viewStart, viewStart + numViews))
    Ptrue = composeP(cfg["A"], W, cfg["k"])
    offs = np.arange(numViews + 1, dtype=np.int64) * N
    model = np.ascontiguousarray(np.tile(corners, (numViews, 1)))
    eng = engine.RefineEngine(cfg["model"], "f64", device)     # ground truth is always fp64
    try:
        eng.setProblem(offs, None, model)
        sensor = eng.evaluate(Ptrue, wantY=True)["y"]
    finally:
        eng.close()
    if noiseSigma > 0:
        rng = np.random.default_rng([noiseSeed, viewStart])
        sensor = sensor + rng.normal(0.0, noiseSigma, sensor.shape)
    L = eng.L
    P0 = Ptrue.copy()
    if perturb:
        P0[:L] *= 1 + perturb * np.random.default_rng(perturbSeed).standard_normal(L)
        ext = P0[L:].reshape(-1, 6)
        # noise is a function of the GLOBAL view index (drawn per block of 1024 views), so a
        # shard sees the same start point whatever the sharding
        blk = 1024
        for b in range(viewStart // blk, (viewStart + numViews - 1) // blk + 1):
            z = np.random.default_rng([perturbSeed + 1, b]).standard_normal((blk, 6))
            lo, hi = max(b * blk, viewStart), min((b + 1) * blk, viewStart + numViews)
            ext[lo - viewStart:hi - viewStart] *= 1 + perturb * z[lo - b * blk:hi - b * blk]
    return dict(viewOffsets=offs, sensorPoints=sensor, modelPoints=model, Ptrue=Ptrue, P0=P0,
                model=cfg["model"], dtype=cfg["dtype"], pointsPerView=N)
