"""Distortion / projection models: the numeric half of the reference's src/distortion.py.

The reference builds its Jacobian by running these formulas on sympy symbols
(src/distortion.py:13-40); here the derivatives are closed-form device code
(csrc/point_model.hpp), so nothing symbolic ships. The forward model itself
(projectWithDistortion / distortPoints) is evaluated by the HIP library too.
"""
import numpy as np

from . import _native as nat
from . import engine


class DistortionModel:
    _maxFOV = 179.5              # src/distortion.py:11 (the NaN-by-FOV branch is disabled there, :12)
    _shouldNaNByFOV = False
    modelName = None
    modelId = None

    def getIntrinsicSymbols(self):
        """names, in parameter-vector order (src/distortion.py:61-62)"""
        return ("α", "β", "γ", "uc", "vc")

    def getDistortionSymbols(self):
        raise NotImplementedError()

    def projectWithDistortion(self, A, X, k):
        """A (3,3), X (N,3) camera-frame points, k -> (N,2) sensor points (src/distortion.py:42-59)"""
        X = np.asarray(X, dtype=np.float64)
        if X.ndim != 2 or X.shape[1] != 3:
            raise ValueError(f"Expected shape (None, 3), got {X.shape}")
        return engine.projectWithDistortion(self.modelId, A, X, self._checkK(k))

    def distortPoints(self, x, k):
        """x (N,2) normalised points -> (N,2) distorted normalised points"""
        x = np.asarray(x, dtype=np.float64)
        if x.ndim != 2 or x.shape[1] != 2:
            raise ValueError(f"Expected shape (None, 2), got {x.shape}")
        return engine.distortPoints(self.modelId, x, self._checkK(k))

    def estimateDistortion(self, A, allDetections, allBoardPosesInCamera, device=0):
        """Linear least-squares start value of k given A and the board poses (src/distortion.py:70,
        110-191 radial-tangential, 222-271 fisheye): D^T D and D^T Ddot are formed on the device
        (calib_distortion_normal_equations), the |k| x |k| system is solved on the host."""
        from . import linearcalibrate
        offs, sensor, model = engine.packDetections(allDetections)
        DtD, Dtd = engine.distortionNormalEquations(self.modelId, offs, sensor, model, A,
                                                    np.asarray(allBoardPosesInCamera, dtype=np.float64), device)
        return linearcalibrate.solveDistortionNormalEquations(DtD, Dtd)

    def _checkK(self, k):
        k = np.asarray(k, dtype=np.float64).ravel()
        n = len(self.getDistortionSymbols())
        if k.shape[0] != n:
            raise ValueError(f"not enough values to unpack (expected {n}, got {k.shape[0]})")
        return k


class RadialTangentialModel(DistortionModel):
    """k = (k1, k2, p1, p2, k3)  (src/distortion.py:74-108)"""
    modelName = "radtan"
    modelId = nat.MODEL_RADTAN

    def getDistortionSymbols(self):
        return ("k1", "k2", "p1", "p2", "k3")


class FisheyeModel(DistortionModel):
    """k = (k1, k2, k3, k4), theta = atan(r) polynomial  (src/distortion.py:194-220).

    At r = 0 the reference evaluates 0/0 (NaN); this engine returns the analytic
    limit (xd, yd) = (0, 0)."""
    modelName = "fisheye"
    modelId = nat.MODEL_FISHEYE

    def getDistortionSymbols(self):
        return ("k1", "k2", "k3", "k4")
