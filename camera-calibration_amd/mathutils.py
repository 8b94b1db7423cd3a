"""Host-side pose helpers used to pack / unpack the parameter vector.

Same names and conventions as the reference's src/mathutils.py (angles in
DEGREES, R = Rz @ Ry @ Rx), vectorised over views: packing a million poses is
O(M) numpy work per refine call, not per LM iteration, and stays on the host
(SURVEY.md section 8(a) row a11).
"""
import numpy as np


def col(v):
    """column vector from a list / tuple (src/mathutils.py:54-56)"""
    return np.array(v).reshape(-1, 1)


def validateShape(inputShape, requiredShape):
    """src/mathutils.py:102-105"""
    for got, want in zip(inputShape, requiredShape):
        if want is not None and want != got:
            raise ValueError(f"Expected shape {requiredShape}, got {inputShape}")


def eulerToRotationMatrices(rhoDegrees):
    """(M,3) Euler angles (rx, ry, rz) in degrees -> (M,3,3).

    Per-axis Rodrigues of the reference's numeric path (src/mathutils.py:36-51,
    59-81): a rotation whose angle satisfies np.isclose(|theta|, 0) is the identity."""
    rho = np.asarray(rhoDegrees, dtype=np.float64).reshape(-1, 3)
    th = np.radians(rho)
    s, c = np.sin(th), np.cos(th)
    tiny = np.isclose(np.abs(th), 0)
    s = np.where(tiny, 0.0, s)
    c = np.where(tiny, 1.0, c)
    sx, sy, sz = s[:, 0], s[:, 1], s[:, 2]
    cx, cy, cz = c[:, 0], c[:, 1], c[:, 2]
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


def eulerToRotationMatrix(rXrYrZDegrees):
    return eulerToRotationMatrices([rXrYrZDegrees])[0]


def rotationMatricesToEuler(R):
    """(M,3,3) -> (M,3) degrees (psi, theta, phi), Slabaugh's decomposition with the
    gimbal-lock branches of src/mathutils.py:13-33."""
    R = np.asarray(R, dtype=np.float64).reshape(-1, 3, 3)
    R31 = R[:, 2, 0]
    lockNeg = np.isclose(R31, -1)
    lockPos = np.isclose(R31, +1)
    regular = ~(lockNeg | lockPos)
    out = np.empty((R.shape[0], 3))
    with np.errstate(invalid="ignore", divide="ignore"):
        th = -np.arcsin(np.clip(R31, -1, 1))
        cth = np.cos(th)
        psi = np.arctan2(R[:, 2, 1] / cth, R[:, 2, 2] / cth)
        phi = np.arctan2(R[:, 1, 0] / cth, R[:, 0, 0] / cth)
    out[:, 0] = np.where(regular, psi, 0.0)
    out[:, 1] = np.where(regular, th, 0.0)
    out[:, 2] = np.where(regular, phi, 0.0)
    if lockNeg.any():
        out[lockNeg, 1] = np.pi / 2
        out[lockNeg, 0] = np.arctan2(R[lockNeg, 0, 1], R[lockNeg, 0, 2])
    if lockPos.any():
        out[lockPos, 1] = -np.pi / 2
        out[lockPos, 0] = np.arctan2(-R[lockPos, 0, 1], -R[lockPos, 0, 2])
    return np.degrees(out)


def rotationMatrixToEuler(R):
    return tuple(rotationMatricesToEuler(np.asarray(R).reshape(1, 3, 3))[0])


def poseFromRT(R, T):
    """src/mathutils.py:140-146"""
    M = np.eye(4)
    M[:3, :3] = R
    M[:3, 3] = np.asarray(T, dtype=np.float64).ravel()
    return M


def posesFromRT(R, T):
    R = np.asarray(R, dtype=np.float64).reshape(-1, 3, 3)
    W = np.zeros((R.shape[0], 4, 4))
    W[:, :3, :3] = R
    W[:, :3, 3] = np.asarray(T, dtype=np.float64).reshape(-1, 3)
    W[:, 3, 3] = 1.0
    return W


def hom(v):
    """src/mathutils.py:120-127"""
    if isinstance(v, (tuple, list)):
        v = np.array(v).reshape(-1, len(v))
    elif v.ndim == 1:
        return np.append(v, 1)
    return np.hstack((v, np.ones((v.shape[0], 1))))


def unhom(vHom):
    """src/mathutils.py:130-137"""
    if vHom.ndim == 1:
        return vHom[:-1] / vHom[-1]
    if vHom.ndim == 2:
        return vHom[:, :-1] / vHom[:, -1:]
    raise ValueError(f"Unexpected input shape for unhom: {vHom.shape}\n{vHom}")


def transform(b_M_a, aX):
    """rigid transform of (N,3) points (src/mathutils.py:195-208)"""
    b_M_a = np.asarray(b_M_a)
    aX = np.asarray(aX)
    validateShape(b_M_a.shape, (4, 4))
    validateShape(aX.shape, (None, 3))
    return aX @ b_M_a[:3, :3].T + b_M_a[:3, 3]


def radians(angleDegrees):
    """src/mathutils.py:9-10"""
    return angleDegrees / 180.0 * np.pi


def skew(v):
    """'hat' operator: (3,1) vector -> skew-symmetric (3,3) (src/mathutils.py:84-94)"""
    v = np.asarray(v)
    validateShape(v.shape, (3, 1))
    x, y, z = v[:, 0]
    return np.array([[0, -z, y], [z, 0, -x], [-y, x, 0]])


def unskew(vHat):
    """inverse of skew: (3,3) -> (3,) (src/mathutils.py:97-99)"""
    vHat = np.asarray(vHat)
    validateShape(vHat.shape, (3, 3))
    return np.array([vHat[2, 1], vHat[0, 2], vHat[1, 0]])


def exp(wHat):
    """so(3) -> SO(3), Rodrigues' formula with the reference's numeric guard: an angle with
    np.isclose(|w|, 0) maps to the identity (src/mathutils.py:59-81). The reference's
    isSymbolic=True branch builds sympy expressions for its lambdified Jacobian; the closed-form
    device Jacobian replaces it (csrc/point_model.hpp), so only the numeric mapping exists here."""
    wHat = np.asarray(wHat, dtype=np.float64)
    w = unskew(wHat)
    n = np.linalg.norm(w)
    if np.isclose(n, 0):
        return np.eye(3)
    K = wHat / n
    return np.eye(3) + np.sin(n) * K + (1.0 - np.cos(n)) * (K @ K)


def stack(A):
    """columns of a matrix stacked into one column (src/mathutils.py:108-110)"""
    return col(np.asarray(A).T.ravel())


def unstack(As):
    """inverse of stack for a square matrix (src/mathutils.py:113-117)"""
    As = np.asarray(As)
    N = int(np.sqrt(As.size))
    return As.reshape((N, N)).T


def normalize(A):
    """scale so that the last element is 1 (src/mathutils.py:137-138)"""
    A = np.asarray(A)
    return A / A.ravel()[-1]


def projectStandard(X):
    """(N,3) camera-frame points -> (N,2) normalised image points (src/mathutils.py:174-192)"""
    X = np.asarray(X, dtype=np.float64)
    validateShape(X.shape, (None, 3))
    return X[:, :2] / X[:, 2:3]


def project(A, wMc, wX):
    """pinhole projection of world points: A (3,3), camera pose in world wMc (4,4), wX (N,3)
    -> (N,2) pixels (src/mathutils.py:149-171)"""
    A = np.asarray(A, dtype=np.float64)
    wMc = np.asarray(wMc, dtype=np.float64)
    wX = np.asarray(wX, dtype=np.float64)
    validateShape(A.shape, (3, 3))
    validateShape(wMc.shape, (4, 4))
    validateShape(wX.shape, (None, 3))
    x = projectStandard(transform(np.linalg.inv(wMc), wX))
    return unhom(hom(x) @ A.T)
