"""ProjectionJacobian drop-in (reference: src/jacobian.py:12-85).

Same constructor and methods; the values come from the closed-form HIP kernel
instead of a sympy-lambdified expression, so construction is instant (the
reference spends 5-25 s in sympy.diff / lambdify, src/calibrate.py:174).
"""
import numpy as np

from . import distortion
from . import engine

DENSE_LIMIT_BYTES = 8 << 30


class ProjectionJacobian:
    _numExtrinsicParamsPerView = 6

    def __init__(self, distortionModel: distortion.DistortionModel, dtype="f64", device=0):
        self._distortionModel = distortionModel
        self._intrinsicAndDistortionSymbols = (distortionModel.getIntrinsicSymbols()
                                               + distortionModel.getDistortionSymbols())
        self._dtype = dtype
        self._device = device
        self._resident = engine.ResidentProblem(distortionModel.modelId, dtype, device)

    def computeCompact(self, P, allModelPoints):
        """(MN, 2, L+6): per point the rows (du, dv) over [shared L | own view's 6] columns.
        This is everything the dense matrix of compute() holds besides structural zeros."""
        offs, model = engine.packModelPoints(allModelPoints)
        eng = self._resident.get(offs, None, model)
        return eng.evaluate(np.asarray(P, dtype=np.float64).ravel(), wantJ=True)["Jc"]

    def compute(self, P, allModelPoints):
        """Dense J, (2*MN, L+6M): rows (u_j, v_j) interleaved per point in view order, columns
        [0,L) shared, [L+6i, L+6i+6) view i, zero elsewhere (src/jacobian.py:62-84)."""
        L = len(self._intrinsicAndDistortionSymbols)
        M = len(allModelPoints)
        MN = sum(np.asarray(m).shape[0] for m in allModelPoints)
        K = L + self._numExtrinsicParamsPerView * M
        if 2 * MN * K * 8 > DENSE_LIMIT_BYTES:
            raise MemoryError(f"dense Jacobian would be {2*MN*K*8/2**30:.1f} GiB "
                              f"({2*MN} x {K}); use computeCompact() (MN,2,{L+6})")
        Jc = self.computeCompact(P, allModelPoints)
        J = np.zeros((2 * MN, K))
        J[:, :L] = Jc[:, :, :L].reshape(2 * MN, L)
        row = 0
        for i, modelPoints in enumerate(allModelPoints):
            n2 = 2 * np.asarray(modelPoints).shape[0]
            c0 = L + 6 * i
            J[row:row + n2, c0:c0 + 6] = Jc[row // 2:(row + n2) // 2, :, L:].reshape(n2, 6)
            row += n2
        return J

    def _blocks(self, intrinsicValues, extrinsicValues, modelPoints):
        P = np.concatenate((np.asarray(intrinsicValues, dtype=np.float64).ravel(),
                            np.asarray(extrinsicValues, dtype=np.float64).ravel()))
        modelPoints = np.asarray(modelPoints, dtype=np.float64)
        Jc = self.computeCompact(P, [modelPoints])
        return Jc.reshape(2 * modelPoints.shape[0], -1)

    def _createIntrinsicsJacobianBlock(self, intrinsicValues, extrinsicValues, modelPoints):
        """(2N, L) block d(u,v)/d(intrinsics, distortion)  (src/jacobian.py:38-41)"""
        L = len(self._intrinsicAndDistortionSymbols)
        return self._blocks(intrinsicValues, extrinsicValues, modelPoints)[:, :L].copy()

    def _createExtrinsicsJacobianBlock(self, intrinsicValues, extrinsicValues, modelPoints):
        """(2N, 6) block d(u,v)/d(rx, ry, rz, tx, ty, tz)  (src/jacobian.py:43-46)"""
        L = len(self._intrinsicAndDistortionSymbols)
        return self._blocks(intrinsicValues, extrinsicValues, modelPoints)[:, L:].copy()


class HomographyJacobian:
    """Drop-in for src/jacobian.py:88-121: same constructor and compute(h, modelPoints); the nine
    closed-form columns come from a device kernel instead of a lambdified sympy expression."""

    def __init__(self, device=0):
        self._device = device

    def compute(self, h, modelPoints):
        """h = (H11 .. H33), modelPoints (N,3) -> J (2N, 9), rows (u_j, v_j) interleaved."""
        return engine.homographyJacobian(h, np.asarray(modelPoints, dtype=np.float64), self._device)


def createJacRadTan() -> ProjectionJacobian:
    """src/jacobian.py:189-192"""
    return ProjectionJacobian(distortion.RadialTangentialModel())
