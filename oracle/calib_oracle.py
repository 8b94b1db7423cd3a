"""CPU oracle: numpy restatement of the reference's LM-refinement hot path.

TEST INFRASTRUCTURE ONLY. Nothing under ``camera-calibration_amd/`` imports this
module; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg
of ``bench.py`` do, and there only as the checker / the reported CPU baseline.

Parity status: PINNED. Every function below is checked by
``tests/test_oracle_golden.py`` against vectors produced by running the
reference itself (``tools/oracle/make_golden.py`` -> ``tests/golden/g*.npz``).

All ``file:line`` citations are relative to the reference checkout
(pvphan/camera-calibration, ``/root/reference``).

The reference differentiates the projection symbolically with sympy and
evaluates the lambdified expression per view (``src/jacobian.py:19-36,147-172``).
This restatement evaluates the same derivatives in closed form (chain rule,
SURVEY.md Appendix A); against the sympy oracle the columns agree to ~1e-15
relative (see the golden test), which is what "restatement" means here.
"""
import numpy as np

DEG = np.pi / 180.0
RADTAN, FISHEYE = 0, 1
MODEL_NAMES = {RADTAN: "radtan", FISHEYE: "fisheye"}
NUM_DISTORTION = {RADTAN: 5, FISHEYE: 4}   # src/distortion.py:75-76,195-196


def numShared(model):
    """L = 5 intrinsics + |k| (src/calibrate.py:246-252)."""
    return 5 + NUM_DISTORTION[model]


# --------------------------------------------------------------------------
# rotations (src/mathutils.py:13-51, 59-81)
# --------------------------------------------------------------------------
def _axisTrig(angleDeg, numericQuirk):
    """sin/cos of one Euler angle given in degrees.

    numericQuirk=True follows the *numeric* Rodrigues path
    (src/mathutils.py:68-80): ``np.radians`` then identity when
    ``np.isclose(|w|, 0)`` i.e. |theta| <= 1e-8 rad. numericQuirk=False follows
    the *symbolic* path the Jacobian is derived from (src/mathutils.py:42-45,
    63-67), which has no such threshold.
    """
    th = np.asarray(angleDeg, dtype=np.float64) * DEG
    s, c = np.sin(th), np.cos(th)
    if numericQuirk:
        tiny = np.abs(th) <= 1e-8
        s = np.where(tiny, 0.0, s)
        c = np.where(tiny, 1.0, c)
    return s, c


def eulerToR(rhoDeg, numericQuirk=True):
    """(M,3) Euler angles in DEGREES -> (M,3,3), R = Rz @ Ry @ Rx (src/mathutils.py:36-51)."""
    rho = np.atleast_2d(np.asarray(rhoDeg, dtype=np.float64))
    sx, cx = _axisTrig(rho[:, 0], numericQuirk)
    sy, cy = _axisTrig(rho[:, 1], numericQuirk)
    sz, cz = _axisTrig(rho[:, 2], numericQuirk)
    R = np.empty((rho.shape[0], 3, 3))
    R[:, 0, 0] = cz * cy
    R[:, 0, 1] = cz * sy * sx - sz * cx
    R[:, 0, 2] = cz * sy * cx + sz * sx
    R[:, 1, 0] = sz * cy
    R[:, 1, 1] = sz * sy * sx + cz * cx
    R[:, 1, 2] = sz * sy * cx - cz * sx
    R[:, 2, 0] = -sy
    R[:, 2, 1] = cy * sx
    R[:, 2, 2] = cy * cx
    return R


def rToEuler(R):
    """(M,3,3) -> (M,3) Euler degrees (psi, theta, phi) = (x, y, z) (src/mathutils.py:13-33)."""
    R = np.asarray(R, dtype=np.float64).reshape(-1, 3, 3)
    out = np.empty((R.shape[0], 3))
    for i, Ri in enumerate(R):
        R11, R12, R13, R21, R22, R23, R31, R32, R33 = Ri.ravel()
        if not (np.isclose(R31, +1) or np.isclose(R31, -1)):
            th = -np.arcsin(R31)
            psi = np.arctan2(R32 / np.cos(th), R33 / np.cos(th))
            phi = np.arctan2(R21 / np.cos(th), R11 / np.cos(th))
        else:
            phi = 0.0
            if np.isclose(R31, -1):
                th = np.pi / 2
                psi = phi + np.arctan2(R12, R13)
            else:
                th = -np.pi / 2
                psi = -phi + np.arctan2(-R12, -R13)
        out[i] = np.degrees((psi, th, phi))
    return out


# --------------------------------------------------------------------------
# parameter vector (src/calibrate.py:199-267)
# --------------------------------------------------------------------------
def composeParameterVector(A, W, k):
    """P = (alpha, beta, gamma, uc, vc, k..., [rx ry rz tx ty tz] per view), shape (K,)."""
    A = np.asarray(A, dtype=np.float64)
    W = np.asarray(W, dtype=np.float64).reshape(-1, 4, 4)
    shared = [A[0, 0], A[1, 1], A[0, 1], A[0, 2], A[1, 2]] + list(k)
    ext = np.hstack((rToEuler(W[:, :3, :3]), W[:, :3, 3]))
    return np.concatenate((np.asarray(shared, dtype=np.float64), ext.ravel()))


def decomposeParameterVector(P, model):
    """-> A (3,3), W (M,4,4), k (|k|,) using the numeric Rodrigues path."""
    P = np.asarray(P, dtype=np.float64).ravel()
    L = numShared(model)
    al, be, ga, uc, vc = P[:5]
    A = np.array([[al, ga, uc], [0, be, vc], [0, 0, 1]], dtype=np.float64)
    ext = P[L:].reshape(-1, 6)
    W = np.tile(np.eye(4), (ext.shape[0], 1, 1))
    W[:, :3, :3] = eulerToR(ext[:, :3], numericQuirk=True)
    W[:, :3, 3] = ext[:, 3:]
    return A, W, P[5:L].copy()


# --------------------------------------------------------------------------
# distortion + projection (src/distortion.py:42-59,78-108,198-220)
# --------------------------------------------------------------------------
def distortPoints(model, x, y, k):
    if model == RADTAN:
        k1, k2, p1, p2, k3 = k
        r = np.sqrt(x * x + y * y)          # np.linalg.norm(x, axis=1)
        rad = 1 + k1 * r**2 + k2 * r**4 + k3 * r**6
        tx = 2 * p1 * x * y + p2 * (r**2 + 2 * x**2)
        ty = p1 * (r**2 + 2 * y**2) + 2 * p2 * x * y
        return rad * x + tx, rad * y + ty
    k1, k2, k3, k4 = k
    r = np.sqrt(x * x + y * y)
    th = np.arctan(r)
    with np.errstate(invalid="ignore", divide="ignore"):
        s = (th / r) * (1 + k1 * th**2 + k2 * th**4 + k3 * th**6 + k4 * th**8)
    return s * x, s * y


def pointViewIndex(viewOffsets):
    viewOffsets = np.asarray(viewOffsets, dtype=np.int64)
    counts = np.diff(viewOffsets)
    return np.repeat(np.arange(counts.shape[0]), counts)


def projectAllPoints(model, P, viewOffsets, modelPoints):
    """Restates Calibrator.projectAllPoints (src/calibrate.py:190-197) -> (MN,2)."""
    P = np.asarray(P, dtype=np.float64).ravel()
    L = numShared(model)
    al, be, ga, uc, vc = P[:5]
    k = P[5:L]
    ext = P[L:].reshape(-1, 6)
    vi = pointViewIndex(viewOffsets)
    R = eulerToR(ext[:, :3], numericQuirk=True)[vi]
    t = ext[:, 3:][vi]
    Pc = np.einsum("nij,nj->ni", R, np.asarray(modelPoints, dtype=np.float64)) + t
    x = Pc[:, 0] / Pc[:, 2]
    y = Pc[:, 1] / Pc[:, 2]
    xd, yd = distortPoints(model, x, y, k)
    return np.stack((al * xd + ga * yd + uc, be * yd + vc), axis=1)


def reprojectionError(model, P, viewOffsets, sensorPoints, modelPoints):
    """src/calibrate.py:178-188: sum over points of squared 2-norm."""
    y = projectAllPoints(model, P, viewOffsets, modelPoints)
    return np.sum(np.linalg.norm(np.asarray(sensorPoints) - y, axis=1) ** 2)


# --------------------------------------------------------------------------
# Jacobian (src/jacobian.py:48-85; closed form per SURVEY Appendix A)
# --------------------------------------------------------------------------
def jacobianCompact(model, P, viewOffsets, modelPoints):
    """-> (MN, 2, C): per point, row 0 = du/d., row 1 = dv/d., columns
    [alpha beta gamma uc vc | k... | rx ry rz tx ty tz] of the point's own view."""
    P = np.asarray(P, dtype=np.float64).ravel()
    L = numShared(model)
    al, be, ga, uc, vc = P[:5]
    k = P[5:L]
    ext = P[L:].reshape(-1, 6)
    vi = pointViewIndex(viewOffsets)
    Pw = np.asarray(modelPoints, dtype=np.float64)
    n = Pw.shape[0]

    sx, cx = _axisTrig(ext[:, 0], False)
    sy, cy = _axisTrig(ext[:, 1], False)
    sz, cz = _axisTrig(ext[:, 2], False)
    R = eulerToR(ext[:, :3], numericQuirk=False)[vi]
    q = np.einsum("nij,nj->ni", R, Pw)
    Pc = q + ext[:, 3:][vi]
    # dPc/drho. = (pi/180) * a. x q with a_x = Rz Ry e_x, a_y = Rz e_y, a_z = e_z
    ax = np.stack((cz * cy, sz * cy, -sy), axis=1)[vi]
    ay = np.stack((-sz, cz, np.zeros_like(sz)), axis=1)[vi]
    az = np.tile(np.array([0.0, 0.0, 1.0]), (n, 1))
    dPc = [DEG * np.cross(a, q) for a in (ax, ay, az)]
    eye = np.eye(3)
    dPc += [np.tile(eye[j], (n, 1)) for j in range(3)]

    iz = 1.0 / Pc[:, 2]
    x = Pc[:, 0] * iz
    y = Pc[:, 1] * iz
    r2 = x * x + y * y

    C = L + 6
    J = np.zeros((n, 2, C))
    if model == RADTAN:
        k1, k2, p1, p2, k3 = k
        rad = 1 + k1 * r2 + k2 * r2**2 + k3 * r2**3
        drad = k1 + 2 * k2 * r2 + 3 * k3 * r2**2
        xd = rad * x + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
        yd = rad * y + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
        xd_x = rad + 2 * x * x * drad + 2 * p1 * y + 6 * p2 * x
        xd_y = 2 * x * y * drad + 2 * p1 * x + 2 * p2 * y
        yd_x = xd_y
        yd_y = rad + 2 * y * y * drad + 6 * p1 * y + 2 * p2 * x
        dxd_dk = [x * r2, x * r2**2, 2 * x * y, r2 + 2 * x * x, x * r2**3]
        dyd_dk = [y * r2, y * r2**2, r2 + 2 * y * y, 2 * x * y, y * r2**3]
    else:
        k1, k2, k3, k4 = k
        r = np.sqrt(r2)
        th = np.arctan(r)
        t2 = th * th
        poly = 1 + k1 * t2 + k2 * t2**2 + k3 * t2**3 + k4 * t2**4
        gp = (1 + 3 * k1 * t2 + 5 * k2 * t2**2 + 7 * k3 * t2**3 + 9 * k4 * t2**4) / (1 + r2)
        with np.errstate(invalid="ignore", divide="ignore"):
            s = th * poly / r
            sr_over_r = (gp * r - th * poly) / (r2 * r)      # s_r / r
            thr = th / r
        small = r < 1e-8        # analytic limit (the reference gives NaN exactly at r = 0)
        s = np.where(small, 1.0, s)
        sr_over_r = np.where(small, 2 * k1 - 2.0 / 3.0, sr_over_r)
        thr = np.where(small, 1.0, thr)
        xd, yd = s * x, s * y
        xd_x = s + x * x * sr_over_r
        xd_y = x * y * sr_over_r
        yd_x = xd_y
        yd_y = s + y * y * sr_over_r
        dxd_dk = [x * thr * t2**j for j in (1, 2, 3, 4)]
        dyd_dk = [y * thr * t2**j for j in (1, 2, 3, 4)]

    # intrinsics (src/distortion.py:57-58: u = alpha xd + gamma yd + uc, v = beta yd + vc)
    J[:, 0, 0] = xd
    J[:, 1, 1] = yd
    J[:, 0, 2] = yd
    J[:, 0, 3] = 1.0
    J[:, 1, 4] = 1.0
    for j, (dx_, dy_) in enumerate(zip(dxd_dk, dyd_dk)):
        J[:, 0, 5 + j] = al * dx_ + ga * dy_
        J[:, 1, 5 + j] = be * dy_
    ux = al * xd_x + ga * yd_x
    uy = al * xd_y + ga * yd_y
    vx = be * yd_x
    vy = be * yd_y
    for j, d in enumerate(dPc):
        dx = (d[:, 0] - x * d[:, 2]) * iz
        dy = (d[:, 1] - y * d[:, 2]) * iz
        J[:, 0, L + j] = ux * dx + uy * dy
        J[:, 1, L + j] = vx * dx + vy * dy
    return J


def jacobianDense(model, P, viewOffsets, modelPoints):
    """ProjectionJacobian.compute layout (src/jacobian.py:62-84): rows (u_j, v_j)
    interleaved, cols [0,L) shared, [L+6i, L+6i+6) view i, zero elsewhere."""
    Jc = jacobianCompact(model, P, viewOffsets, modelPoints)
    L = numShared(model)
    vi = pointViewIndex(viewOffsets)
    M = len(viewOffsets) - 1
    n = Jc.shape[0]
    J = np.zeros((2 * n, L + 6 * M))
    J[:, :L] = Jc[:, :, :L].reshape(2 * n, L)
    rows = np.arange(2 * n)
    for j in range(6):
        J[rows, L + 6 * np.repeat(vi, 2) + j] = Jc[:, :, L + j].reshape(2 * n)
    return J


# --------------------------------------------------------------------------
# LM, reference-exact dense form (src/calibrate.py:117-171)
# --------------------------------------------------------------------------
LAMBDA_INITIAL, LAMBDA_MIN, LAMBDA_MAX, PT_ERROR_MIN = 1e-3, 1e-10, 1e10, 1e-12


def refineDense(model, P0, viewOffsets, sensorPoints, modelPoints, maxIters,
                lamInit=LAMBDA_INITIAL, lamMin=LAMBDA_MIN, lamMax=LAMBDA_MAX,
                errMin=PT_ERROR_MIN):
    """Dense J, dense J^T J, explicit inv: the reference loop line by line.
    Returns (Pt_error [pre-update, as the reference returns it], P, trace) with
    trace rows (iter, err_P, err_P1, lambda, accepted)."""
    Pt = np.array(P0, dtype=np.float64).reshape(-1, 1)
    ydot = np.asarray(sensorPoints, dtype=np.float64)
    lam = lamInit
    trace = []
    Pt_error = None
    for it in range(maxIters):
        J = jacobianDense(model, Pt, viewOffsets, modelPoints)
        JTJ = J.T @ J
        diagJTJ = np.diag(np.diagonal(JTJ))
        y = projectAllPoints(model, Pt, viewOffsets, modelPoints)
        r = ydot.reshape(-1, 1) - y.reshape(-1, 1)
        delta = np.linalg.inv(JTJ + lam * diagJTJ) @ J.T @ r
        Pt_error = reprojectionError(model, Pt, viewOffsets, ydot, modelPoints)
        Pt1_error = reprojectionError(model, Pt + delta, viewOffsets, ydot, modelPoints)
        accepted = bool(Pt1_error < Pt_error)
        trace.append((it, Pt_error, Pt1_error, lam, float(accepted)))
        if accepted:
            Pt = Pt + delta
            lam /= 10
        else:
            lam *= 10
        if not (lamMin < lam < lamMax) or Pt_error < errMin:
            break
    if Pt_error is None:
        raise UnboundLocalError("maxIters=0: Pt_error referenced before assignment "
                                "(src/calibrate.py:171)")
    return Pt_error, Pt.ravel(), np.array(trace)


# --------------------------------------------------------------------------
# block-arrow normal equations + Schur step (equals the
# dense step of src/calibrate.py:146-152 to ~1e-11)
# --------------------------------------------------------------------------
def normalBlocks(model, Jc, r, viewOffsets):
    """Jc (MN,2,C), r (MN,2) -> B (L,L), E (M,L,6), V (M,6,6), g (K,)."""
    L = numShared(model)
    viewOffsets = np.asarray(viewOffsets, dtype=np.int64)
    M = viewOffsets.shape[0] - 1
    n = Jc.shape[0]
    Jr = Jc.reshape(2 * n, -1)
    rr = np.asarray(r, dtype=np.float64).reshape(2 * n)
    B = Jr[:, :L].T @ Jr[:, :L]
    g = np.empty(L + 6 * M)
    g[:L] = Jr[:, :L].T @ rr
    E = np.empty((M, L, 6))
    V = np.empty((M, 6, 6))
    for i in range(M):
        a, b = 2 * viewOffsets[i], 2 * viewOffsets[i + 1]
        Ji, Je = Jr[a:b, :L], Jr[a:b, L:]
        E[i] = Ji.T @ Je
        V[i] = Je.T @ Je
        g[L + 6 * i:L + 6 * i + 6] = Je.T @ rr[a:b]
    return B, E, V, g


def schurStep(B, E, V, g, lam):
    """Solve (J^T J + lam diag(J^T J)) delta = J^T r by eliminating the 6x6 view blocks."""
    L = B.shape[0]
    M = V.shape[0]
    S = B + lam * np.diag(np.diagonal(B))
    s = g[:L].copy()
    Vh = V + lam * np.einsum("mii->mi", V)[:, :, None] * np.eye(6)
    gv = g[L:].reshape(M, 6)
    VinvEt = np.linalg.solve(Vh, np.transpose(E, (0, 2, 1)))      # (M,6,L)
    Vinvg = np.linalg.solve(Vh, gv[:, :, None])[:, :, 0]           # (M,6)
    S = S - np.einsum("mlj,mjk->lk", E, VinvEt)
    s = s - np.einsum("mlj,mj->l", E, Vinvg)
    dc = np.linalg.solve(S, s)
    dv = Vinvg - np.einsum("mjl,l->mj", VinvEt, dc)
    return np.concatenate((dc, dv.ravel()))


def lmStepSchur(model, P, viewOffsets, sensorPoints, modelPoints, lam):
    Jc = jacobianCompact(model, P, viewOffsets, modelPoints)
    r = np.asarray(sensorPoints, dtype=np.float64) - projectAllPoints(
        model, P, viewOffsets, modelPoints)
    B, E, V, g = normalBlocks(model, Jc, r, viewOffsets)
    return schurStep(B, E, V, g, lam)


def refineSchur(model, P0, viewOffsets, sensorPoints, modelPoints, maxIters,
                lamInit=LAMBDA_INITIAL, lamMin=LAMBDA_MIN, lamMax=LAMBDA_MAX,
                errMin=PT_ERROR_MIN):
    """Same loop as refineDense with the step solved through the Schur complement
    (what the device implements). Same return convention."""
    Pt = np.array(P0, dtype=np.float64).ravel()
    ydot = np.asarray(sensorPoints, dtype=np.float64)
    lam = lamInit
    trace = []
    Pt_error = None
    for it in range(maxIters):
        delta = lmStepSchur(model, Pt, viewOffsets, ydot, modelPoints, lam)
        Pt_error = reprojectionError(model, Pt, viewOffsets, ydot, modelPoints)
        Pt1_error = reprojectionError(model, Pt + delta, viewOffsets, ydot, modelPoints)
        accepted = bool(Pt1_error < Pt_error)
        trace.append((it, Pt_error, Pt1_error, lam, float(accepted)))
        if accepted:
            Pt = Pt + delta
            lam /= 10
        else:
            lam *= 10
        if not (lamMin < lam < lamMax) or Pt_error < errMin:
            break
    if Pt_error is None:
        raise UnboundLocalError("maxIters=0 (src/calibrate.py:171)")
    return Pt_error, Pt, np.array(trace)


# --------------------------------------------------------------------------
# synthetic poses (src/dataset.py:59-95, src/checkerboard.py:9-17)
# --------------------------------------------------------------------------
def checkerboardCorners(numW, numH, spacing):
    ii, jj = np.meshgrid(np.arange(numW), np.arange(numH))
    return np.stack((ii.ravel() * spacing, jj.ravel() * spacing,
                     np.zeros(numW * numH)), axis=1).astype(np.float64)


def syntheticBoardPoses(corners, viewIndices):
    """board pose in camera for each (global) view index: legacy RNG seeded per view
    (src/dataset.py:64-70), camera pose composed as src/dataset.py:84-95, inverted (:76)."""
    out = np.empty((len(viewIndices), 4, 4))
    Rflip = eulerToR([[180.0, 0.0, 0.0]])[0]
    for j, vi in enumerate(viewIndices):
        rs = np.random.RandomState(int(vi))
        ci = rs.choice(corners.shape[0])
        rx = rs.uniform(-30, 30)
        ry = rs.uniform(-30, 30)
        rz = rs.uniform(-180, 180)
        d = rs.uniform(0.5, 1.0)
        Ma = np.eye(4); Ma[:3, :3] = Rflip; Ma[:3, 3] = corners[ci]
        Mb = np.eye(4); Mb[:3, :3] = eulerToR([[rx, ry, rz]])[0]
        Mc = np.eye(4); Mc[:3, 3] = (0, 0, -d)
        out[j] = np.linalg.inv(Ma @ Mb @ Mc)
    return out
