"""Zhang's closed-form initialisation, on the host (reference: src/linearcalibrate.py,
src/calibrate.py:41-115, src/distortion.py:110-191,222-271).

BASELINE.json keeps this stage on the host; it runs once per calibration, before the device
refinement. Unlike the reference's per-point / per-view Python loops it is written batched over
views (numpy's stacked SVD / solve), so it stays usable at 10^4..10^6 views:
  estimateHomographies   normalised DLT, one (2N x 9) SVD per view      linearcalibrate.py:24-58
  refineHomographies     9-parameter LM per view, 20 iterations          calibrate.py:60-111
  computeIntrinsicMatrix V b = 0 by SVD, A from the Cholesky factor of B  linearcalibrate.py:93-158,266-303
  computeExtrinsics      [r0 r1 t] = A^-1 H / lambda, SVD projection on SO(3)   linearcalibrate.py:306-371
  estimateDistortion     linear least squares for k                      distortion.py:110-191,222-271
"""
import numpy as np

from . import mathutils as mu

_LAMBDA_MIN, _LAMBDA_MAX, _ERROR_MIN = 1e-10, 1e+10, 1e-12      # src/calibrate.py:14-16
_SVD_POINT_LIMIT = 100_000      # stacked (2N x 9) SVDs up to this many points per batch, eigh(M^T M) beyond


def _groupByCount(allDetections):
    """indices of views grouped by their number of points (stacked linear algebra needs equal shapes)"""
    groups = {}
    for i, (Xa, Xb) in enumerate(allDetections):
        groups.setdefault(np.asarray(Xa).shape[0], []).append(i)
    return groups


def computeNormalizationMatrices(X):
    """X (B,N,2) -> (B,3,3): centroid to the origin, mean distance sqrt(2)
    (Hartley & Zisserman 4.4.4; src/linearcalibrate.py:61-90)"""
    mean = X.mean(axis=1)
    dist = np.linalg.norm(X - mean[:, None, :], axis=2).mean(axis=1)
    s = np.sqrt(2) / dist
    M = np.zeros((X.shape[0], 3, 3))
    M[:, 0, 0] = s
    M[:, 1, 1] = s
    M[:, 0, 2] = -s * mean[:, 0]
    M[:, 1, 2] = -s * mean[:, 1]
    M[:, 2, 2] = 1.0
    return M


def _estimateHomographyBatch(Xa, Xb):
    """Xa (B,N,2) sensor, Xb (B,N,2) model -> (B,3,3), H[2,2] = 1"""
    Na, Nb = computeNormalizationMatrices(Xa), computeNormalizationMatrices(Xb)
    a = Xa * Na[:, None, [0, 1], [0, 1]] + Na[:, None, :2, 2]
    b = Xb * Nb[:, None, [0, 1], [0, 1]] + Nb[:, None, :2, 2]
    u, v, X, Y = a[..., 0], a[..., 1], b[..., 0], b[..., 1]
    B, N = u.shape
    if B * N <= _SVD_POINT_LIMIT:
        # the reference's route: right singular vector of M's smallest singular value
        M = np.zeros((B, 2 * N, 9))
        M[:, 0::2, 0], M[:, 0::2, 1], M[:, 0::2, 2] = -X, -Y, -1.0
        M[:, 0::2, 6], M[:, 0::2, 7], M[:, 0::2, 8] = u * X, u * Y, u
        M[:, 1::2, 3], M[:, 1::2, 4], M[:, 1::2, 5] = -X, -Y, -1.0
        M[:, 1::2, 6], M[:, 1::2, 7], M[:, 1::2, 8] = v * X, v * Y, v
        Vt = np.linalg.svd(M, full_matrices=False)[2]
        Hp = Vt[:, -1, :].reshape(B, 3, 3)
    else:
        # the same vector as the eigenvector of M^T M of the smallest eigenvalue, with M^T M built
        # from its 3x3 blocks (p = (X, Y, 1)): [[S, 0, -Su], [0, S, -Sv], [., ., Suu+vv]] -- no 2N x 9
        # matrices, ~50x faster; the normalisation keeps cond(M) ~ 1e2, so squaring it costs
        # nothing that the LM polish of the homographies does not remove
        Pm = np.stack((X, Y, np.ones_like(X)), axis=2)
        S0 = np.einsum("bni,bnj->bij", Pm, Pm)
        Su = np.einsum("bni,bnj->bij", Pm * u[..., None], Pm)
        Sv = np.einsum("bni,bnj->bij", Pm * v[..., None], Pm)
        Se = np.einsum("bni,bnj->bij", Pm * (u * u + v * v)[..., None], Pm)
        MtM = np.zeros((B, 9, 9))
        MtM[:, 0:3, 0:3] = S0
        MtM[:, 3:6, 3:6] = S0
        MtM[:, 0:3, 6:9] = -Su
        MtM[:, 6:9, 0:3] = -Su
        MtM[:, 3:6, 6:9] = -Sv
        MtM[:, 6:9, 3:6] = -Sv
        MtM[:, 6:9, 6:9] = Se
        Hp = np.linalg.eigh(MtM)[1][:, :, 0].reshape(B, 3, 3)
    H = np.linalg.inv(Na) @ Hp @ Nb
    return H / H[:, 2:3, 2:3]


def estimateHomographies(allDetections):
    """list of (sensor (N,2), model (N,3)) -> list of (3,3) model-plane -> sensor homographies"""
    Hs = [None] * len(allDetections)
    for n, idx in _groupByCount(allDetections).items():
        Xa = np.stack([np.asarray(allDetections[i][0], dtype=np.float64)[:, :2] for i in idx])
        Xb = np.stack([np.asarray(allDetections[i][1], dtype=np.float64)[:, :2] for i in idx])
        mu.validateShape(Xa.shape[1:], (None, 2))
        for i, H in zip(idx, _estimateHomographyBatch(Xa, Xb)):
            Hs[i] = H
    return Hs


def _projectHomography(h, XY):
    """h (B,9), XY (B,N,2) -> projected (B,N,2) and the denominators"""
    X, Y = XY[..., 0], XY[..., 1]
    w = h[:, 6:7] * X + h[:, 7:8] * Y + h[:, 8:9]
    u = (h[:, 0:1] * X + h[:, 1:2] * Y + h[:, 2:3]) / w
    v = (h[:, 3:4] * X + h[:, 4:5] * Y + h[:, 5:6]) / w
    return np.stack((u, v), axis=2), w


def _refineHomographyBatch(H, Xa, Xb, maxIters=20):
    """Levenberg-Marquardt on the 9 entries of each homography (src/calibrate.py:69-111): lambda
    starts at 1e-3, /10 on an accepted step, x10 otherwise, per view; a view stops when lambda
    leaves (1e-10, 1e10) or its error drops below 1e-12."""
    B, N = Xa.shape[0], Xa.shape[1]
    Pt = H.reshape(B, 9).copy()
    lam = np.full(B, 1e-3)
    active = np.ones(B, dtype=bool)
    X, Y = Xb[..., 0], Xb[..., 1]
    ones = np.ones_like(X)
    for _ in range(maxIters):
        if not active.any():
            break
        y, w = _projectHomography(Pt, Xb)
        iw = 1.0 / w
        p = np.stack((X * iw, Y * iw, ones * iw), axis=2)                 # d(u or v)/d(its own row)
        J = np.zeros((B, 2 * N, 9))
        J[:, 0::2, 0:3] = p
        J[:, 0::2, 6:9] = -y[..., 0:1] * p
        J[:, 1::2, 3:6] = p
        J[:, 1::2, 6:9] = -y[..., 1:2] * p
        r = (Xa - y).reshape(B, 2 * N)
        JTJ = np.einsum("bni,bnj->bij", J, J)
        diag = np.einsum("bii->bi", JTJ)
        damped = JTJ + lam[:, None, None] * (diag[:, :, None] * np.eye(9))
        g = np.einsum("bni,bn->bi", J, r)
        with np.errstate(all="ignore"):
            try:
                delta = np.linalg.solve(damped, g[:, :, None])[:, :, 0]
            except np.linalg.LinAlgError:
                delta = np.stack([np.linalg.lstsq(d, gg, rcond=None)[0] for d, gg in zip(damped, g)])
        err0 = np.sum((Xa - y) ** 2, axis=(1, 2))
        y1, _ = _projectHomography(Pt + delta, Xb)
        err1 = np.sum((Xa - y1) ** 2, axis=(1, 2))
        accept = active & (err1 < err0)
        Pt[accept] += delta[accept]
        lam = np.where(active, np.where(accept, lam / 10, lam * 10), lam)
        active &= (lam > _LAMBDA_MIN) & (lam < _LAMBDA_MAX) & ~(err0 < _ERROR_MIN)
    Href = Pt.reshape(B, 3, 3)
    return Href / Href[:, 2:3, 2:3]


def refineHomographies(Hs, allDetections):
    out = [None] * len(Hs)
    for n, idx in _groupByCount(allDetections).items():
        Xa = np.stack([np.asarray(allDetections[i][0], dtype=np.float64)[:, :2] for i in idx])
        Xb = np.stack([np.asarray(allDetections[i][1], dtype=np.float64)[:, :2] for i in idx])
        H = np.stack([Hs[i] for i in idx])
        for i, Hr in zip(idx, _refineHomographyBatch(H, Xa, Xb)):
            out[i] = Hr
    return out


def vecHomography(H, p, q):
    """Burger eq. 96 (src/linearcalibrate.py:161-189); H (...,3,3) -> (...,6)"""
    H = np.asarray(H)
    return np.stack((
        H[..., 0, p] * H[..., 0, q],
        H[..., 0, p] * H[..., 1, q] + H[..., 1, p] * H[..., 0, q],
        H[..., 1, p] * H[..., 1, q],
        H[..., 2, p] * H[..., 0, q] + H[..., 0, p] * H[..., 2, q],
        H[..., 2, p] * H[..., 1, q] + H[..., 1, p] * H[..., 2, q],
        H[..., 2, p] * H[..., 2, q]), axis=-1)


def computeIntrinsicMatrixFrombCholesky(b):
    """B = (A^-1)^T A^-1 = L L^T  =>  A = (L^T)^-1, scaled to A[2,2] = 1 (src/linearcalibrate.py:266-303)"""
    B0, B1, B2, B3, B4, B5 = b
    sign = -1.0 if (B0 < 0 or B2 < 0 or B5 < 0) else 1.0
    Bm = sign * np.array([[B0, B1, B3], [B1, B2, B4], [B3, B4, B5]])
    Lc = np.linalg.cholesky(Bm)
    A = np.linalg.inv(Lc.T)
    return A / A[2, 2]


def computeIntrinsicMatrix(Hs):
    """src/linearcalibrate.py:93-158: stack (v01, v00 - v11) per homography, null vector by SVD"""
    H = np.asarray(Hs, dtype=np.float64).reshape(-1, 3, 3)
    V = np.empty((2 * H.shape[0], 6))
    V[0::2] = vecHomography(H, 0, 1)
    V[1::2] = vecHomography(H, 0, 0) - vecHomography(H, 1, 1)
    b = np.linalg.svd(V, full_matrices=False)[2][-1]
    A = computeIntrinsicMatrixFrombCholesky(tuple(b))
    if np.isnan(A).any():
        raise ValueError(f"Computed intrinsic matrix contains NaN: \n{A}")
    return A


def computeExtrinsics(Hs, A):
    """world-to-camera poses from the homographies (src/linearcalibrate.py:306-371): columns
    r0, r1, t = A^-1 h / |A^-1 h0|, r2 = r0 x r1, nearest rotation by SVD (Zhang, appendix C)"""
    H = np.asarray(Hs, dtype=np.float64).reshape(-1, 3, 3)
    Q = np.linalg.inv(A) @ H                                   # columns A^-1 h0, h1, h2
    lam = np.linalg.norm(Q[:, :, 0], axis=1)
    Q = Q / lam[:, None, None]
    r0, r1, t = Q[:, :, 0], Q[:, :, 1], Q[:, :, 2]
    Qr = np.stack((r0, r1, np.cross(r0, r1)), axis=2)
    U, _, Vt = np.linalg.svd(Qr)
    return list(mu.posesFromRT(U @ Vt, t))


def estimateDistortion(distortionModel, A, allDetections, allBoardPosesInCamera):
    """Linear least squares for the distortion coefficients given A and the poses
    (src/distortion.py:110-191 radial-tangential, :222-271 fisheye -- the latter reproduces the
    reference's formulation, which its author flags as unreliable, tests/test_distortion.py:152).
    Built for all points at once; rows (u_j, v_j) interleaved in view order as in the reference."""
    A = np.asarray(A, dtype=np.float64)
    fx, fy, uc, vc = A[0, 0], A[1, 1], A[0, 2], A[1, 2]
    counts = np.array([np.asarray(s).shape[0] for s, m in allDetections], dtype=np.int64)
    Udot = np.vstack([np.asarray(s, dtype=np.float64).reshape(-1, 2) for s, m in allDetections])
    bX = np.vstack([np.asarray(m, dtype=np.float64).reshape(-1, 3) for s, m in allDetections])
    W = np.asarray(allBoardPosesInCamera, dtype=np.float64).reshape(-1, 4, 4)
    vi = np.repeat(np.arange(counts.shape[0]), counts)
    c = np.einsum("nij,nj->ni", W[vi, :3, :3], bX) + W[vi, :3, 3]
    xn, yn = c[:, 0] / c[:, 2], c[:, 1] / c[:, 2]
    r = np.sqrt(xn * xn + yn * yn)
    # undistorted projection, unhom(A @ hom(x)) (src/mathutils.py:153-171)
    u = fx * xn + A[0, 1] * yn + uc
    v = fy * yn + vc
    if distortionModel.modelName == "radtan":
        Du = np.stack(((u - uc) * r**2, (u - uc) * r**4, fx * (2 * xn * yn),
                       fx * (r**2 + 2 * xn**2), (u - uc) * r**6), axis=1)
        Dv = np.stack(((v - vc) * r**2, (v - vc) * r**4, fy * (r**2 + 2 * yn**2),
                       fy * (2 * xn * yn), (v - vc) * r**6), axis=1)
    else:
        th = np.arctan(r)
        with np.errstate(invalid="ignore", divide="ignore"):
            tr = th / r
        Du = np.stack([fx * (u - uc) * tr * th**(2 * j) for j in (1, 2, 3, 4)], axis=1)
        Dv = np.stack([fy * (v - vc) * tr * th**(2 * j) for j in (1, 2, 3, 4)], axis=1)
    D = np.empty((2 * Du.shape[0], Du.shape[1]))
    D[0::2], D[1::2] = Du, Dv
    Ddot = np.empty(2 * Du.shape[0])
    Ddot[0::2], Ddot[1::2] = Udot[:, 0] - u, Udot[:, 1] - v
    if D.shape[0] <= 2 * _SVD_POINT_LIMIT:
        k = np.linalg.pinv(D) @ Ddot                       # the reference's route
    else:
        k = np.linalg.solve(D.T @ D, D.T @ Ddot)           # |k| x |k| normal equations at scale
    return tuple(k.ravel())


def solveDistortionNormalEquations(DtD, Dtd):
    """k from D^T D k = D^T Ddot (what pinv(D) @ Ddot gives for a full-rank D); the columns of D span
    many orders of magnitude (r^2 ... r^6), so the system is equilibrated before it is solved."""
    DtD, Dtd = np.asarray(DtD, dtype=np.float64), np.asarray(Dtd, dtype=np.float64)
    scale = 1.0 / np.sqrt(np.where(np.diagonal(DtD) > 0, np.diagonal(DtD), 1.0))
    Gs = DtD * scale[:, None] * scale[None, :]
    return tuple(scale * np.linalg.lstsq(Gs, scale * Dtd, rcond=None)[0])


def estimateCalibrationParametersDevice(distortionModel, viewOffsets, sensorPoints, modelPoints, device=0):
    """src/calibrate.py:41-58 with every per-view / per-point stage on the device (DLT + LM polish of
    the homographies, extrinsics, the distortion normal equations); the 6-unknown intrinsics fit
    and the <= 5-unknown distortion solve are host numpy. Packed (CSR) correspondences in,
    (Ainitial, Winitial (M,4,4), kInitial) out."""
    from . import engine
    Hsref = engine.estimateHomographies(viewOffsets, sensorPoints, modelPoints, 20, device)
    Ainitial = computeIntrinsicMatrix(Hsref)
    Winitial = engine.computeExtrinsics(Hsref, Ainitial, device)
    DtD, Dtd = engine.distortionNormalEquations(distortionModel.modelId, viewOffsets, sensorPoints, modelPoints,
                                                Ainitial, Winitial, device)
    return Ainitial, Winitial, solveDistortionNormalEquations(DtD, Dtd)


def estimateCalibrationParameters(distortionModel, allDetections, refine=None):
    """src/calibrate.py:41-58 -> (Ainitial, Winitial, kInitial). `refine(Hs, allDetections)` polishes the
    homographies (default: the batched host implementation above; Calibrator passes the device kernel)."""
    Hs = estimateHomographies(allDetections)
    Hsref = (refine or refineHomographies)(Hs, allDetections)
    Ainitial = computeIntrinsicMatrix(Hsref)
    Winitial = computeExtrinsics(Hsref, Ainitial)
    kInitial = estimateDistortion(distortionModel, Ainitial, allDetections, Winitial)
    return Ainitial, Winitial, kInitial
