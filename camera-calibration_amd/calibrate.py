"""Calibrator drop-in for the nonlinear stage (reference: src/calibrate.py:12-282).

`refineCalibrationParameters`, `projectAllPoints`, `_computeReprojectionError`
and the parameter (de)composition keep the reference's names, argument order
and return types; the LM loop itself runs on the MI355X through the C-ABI of
include/calib_lm.h. The DLT / closed-form initialisation
(`estimateCalibrationParameters`) stays host code (linearcalibrate.py).
"""
import operator
import time

import numpy as np

from . import distortion
from . import engine
from . import jacobian
from . import mathutils as mu


class Calibrator:
    _λinitial = engine.LAMBDA_INITIAL       # src/calibrate.py:13-16
    _λmin = engine.LAMBDA_MIN
    _λmax = engine.LAMBDA_MAX
    _Pt_error_min = engine.PT_ERROR_MIN

    def __init__(self, distortionModel: distortion.DistortionModel, *, dtype="f64", device=0):
        self._distortionModel = distortionModel
        self._jac = None
        self._dtype = dtype
        self._device = device
        self._resident = engine.ResidentProblem(distortionModel.modelId, dtype, device)
        self.lastTrace = None        # (iters, 5+L) rows of the last refine (see calib_lm.h)
        self.lastSeconds = {}        # host time of the last refine call by stage: pack / compose / compare / upload / lm / decompose

    def close(self):
        """release the resident engine (also happens when the Calibrator is collected)"""
        self._resident.close()

    # ---- full pipeline (host initialisation + device refinement) -------------------------
    def calibrate(self, allDetections, maxIters):
        """src/calibrate.py:21-39 -> (sse, Afinal, Wfinal, kFinal)"""
        Ainitial, Winitial, kInitial = self.estimateCalibrationParameters(allDetections)
        return self.refineCalibrationParameters(Ainitial, Winitial, kInitial, allDetections,
                                                maxIters, shouldPrint=True)

    def estimateCalibrationParameters(self, allDetections):
        """src/calibrate.py:41-58: Zhang's closed-form initialisation, on the host."""
        from . import linearcalibrate
        offs, sensor, model = engine.packDetections(allDetections)
        A, W, k = linearcalibrate.estimateCalibrationParametersDevice(self._distortionModel, offs, sensor, model,
                                                                      self._device)
        return A, list(W), k

    def _refineHomographies(self, Hs, allDetections):
        """LM polish of the DLT homographies, all views in one device launch (src/calibrate.py:60-67)"""
        offs, sensor, model = engine.packDetections(allDetections)
        return list(engine.refineHomographies(Hs, offs, sensor, model, 20, self._device))

    def _refineHomography(self, H, sensorPoints, modelPoints, jac=None):
        """LM polish of ONE homography (src/calibrate.py:69-111): the same 20-iteration loop, on the
        device; `jac` (a HomographyJacobian in the reference) is accepted and not needed."""
        s = np.asarray(sensorPoints, dtype=np.float64).reshape(-1, 2)
        m = np.asarray(modelPoints, dtype=np.float64).reshape(-1, 3)
        offs = np.array([0, s.shape[0]], dtype=np.int64)
        return engine.refineHomographies(np.asarray(H, dtype=np.float64).reshape(1, 3, 3), offs, s, m, 20,
                                         self._device)[0]

    def _projectPointsHomography(self, H, modelPoints):
        """src/calibrate.py:113-115"""
        modelPoints = np.asarray(modelPoints, dtype=np.float64)
        return mu.unhom((np.asarray(H, dtype=np.float64) @ mu.hom(modelPoints[:, :2]).T).T)

    # ---- the hot path --------------------------------------------------------------------
    def refineCalibrationParameters(self, Ainitial, Winitial, kInitial, allDetections,
                                    maxIters, shouldPrint=False):
        """Levenberg-Marquardt over all parameters (src/calibrate.py:117-171).

        Returns (Pt_error, Arefined, Wrefined, kRefined) where Pt_error is, as in the
        reference, the error evaluated BEFORE the last update."""
        self._initializeJacobian()
        if not isinstance(self._jac, jacobian.ProjectionJacobian):
            # the reference's loop calls self._jac.compute(Pt, allModelPoints) every iteration (src/calibrate.py:144)
            # and its tests put a mock there (tests/test_calibrate.py:85-90); here the Jacobian blocks are formed
            # inside the device kernel and never pass through that seam, so an injected object cannot be honoured
            raise TypeError(f"Calibrator._jac was replaced by {type(self._jac).__name__}: the device refinement "
                            "evaluates the closed-form projection Jacobian inside its kernels and cannot consult an "
                            "injected Jacobian object; use jacobian.ProjectionJacobian (or leave _jac unset)")
        maxIters = operator.index(maxIters)
        if maxIters <= 0:
            raise UnboundLocalError("local variable 'Pt_error' referenced before assignment")
        t0 = time.perf_counter()
        Pt = self._composeParameterVector(Ainitial, Winitial, kInitial)
        t1 = time.perf_counter()
        # the list goes to the device as it is: no np.vstack of the views on the host (src/calibrate.py:277-282 stacks them
        # per call; here the staged upload gathers from the per-view arrays, engine.ResidentProblem.getFromDetections)
        eng = self._resident.getFromDetections(allDetections)
        sse, P, iters, trace = self._refineOn(eng, Pt, maxIters, shouldPrint)
        t3 = time.perf_counter()
        Arefined, Wrefined, kRefined = self._decomposeParameterVector(P)
        self.lastSeconds.update(compose=t1 - t0, decompose=time.perf_counter() - t3)
        return sse, Arefined, Wrefined, kRefined

    def refinePacked(self, P0, viewOffsets, sensorPoints, modelPoints, maxIters, shouldPrint=False, sameProblem=False):
        """Same loop on already-stacked correspondences (CSR over views): the form that scales
        to millions of views. -> (sse, P (K,), iters, trace)
        sameProblem=True: the caller vouches that these are the (unchanged) arrays of the previous call on this
        Calibrator -- no compare, no upload (engine.ResidentProblem); by default small problems are recognised by
        content and large ones are uploaded again."""
        if maxIters <= 0:
            raise UnboundLocalError("local variable 'Pt_error' referenced before assignment")
        eng = self._resident.get(viewOffsets, sensorPoints, modelPoints, sameProblem=sameProblem)
        return self._refineOn(eng, P0, maxIters, shouldPrint)

    def _refineOn(self, eng, P0, maxIters, shouldPrint):
        t0 = time.perf_counter()
        if not shouldPrint:
            out = eng.refine(P0, maxIters, self._λinitial, self._λmin, self._λmax, self._Pt_error_min)
        else:
            out = self._refineVerbose(eng, P0, maxIters)
        self.lastSeconds = dict(self._resident.lastSeconds, lm=time.perf_counter() - t0, iters=int(out[2]))
        self.lastTrace = out[3]
        return out

    def _refineVerbose(self, eng, P0, maxIters):
        # one round at a time so that the per-iteration print of src/calibrate.py:158-159
        # appears while the loop runs, with real elapsed times
        ts = time.time()
        eng.lmBegin(P0, maxIters, self._λinitial, self._λmin, self._λmax, self._Pt_error_min)
        eng.lmRun(1)
        for it in range(maxIters):
            eng.lmRun(1)
            row, itersDone = eng.peekTrace(it)
            if itersDone > it:
                error = row[1] if not (row[2] < row[1]) else row[2]     # min(Pt1_error, Pt_error)
                self._printIterationStats(it, ts, row[5:], error, row[3])
            if eng.lmDone():
                break
        return eng.lmEnd()

    def _initializeJacobian(self):
        # src/calibrate.py:173-176; instantaneous here (no symbolic differentiation)
        if self._jac is None:
            self._jac = jacobian.ProjectionJacobian(self._distortionModel, self._dtype, self._device)

    def _computeReprojectionError(self, P, allDetections):
        """sum over points of ||sensor - projection||^2 (src/calibrate.py:178-183)"""
        eng = self._resident.getFromDetections(allDetections)
        return eng.evaluate(np.asarray(P, dtype=np.float64).ravel())["sse"]

    def _computeTotalError(self, ydot, y):
        """src/calibrate.py:185-188"""
        return np.sum(np.linalg.norm(np.asarray(ydot) - np.asarray(y), axis=1) ** 2)

    def projectAllPoints(self, P, allModelPoints):
        """(MN,2) projection of every view's model points (src/calibrate.py:190-197)"""
        offs, model = engine.packModelPoints(allModelPoints)
        eng = self._resident.get(offs, None, model)
        return eng.evaluate(np.asarray(P, dtype=np.float64).ravel(), wantY=True)["y"]

    # ---- parameter vector: per-view Euler (de)composition on the device -------------------
    def _composeParameterVector(self, A, W, k):
        """P = (α, β, γ, uc, vc, k..., [ρx, ρy, ρz, tx, ty, tz] per view)^T, shape (K,1);
        rotations as Euler angles in degrees (src/calibrate.py:199-229), calib_compose_params."""
        return engine.composeParameters(self._distortionModel.modelId, A, W, k, self._device).reshape(-1, 1)

    def _decomposeParameterVector(self, P):
        """-> A (3,3), W list of (4,4), k (|k|,)  (src/calibrate.py:231-267), calib_decompose_params"""
        A, W, k = engine.decomposeParameters(self._distortionModel.modelId, P, self._device)
        return A, list(W), k

    def _printIterationStats(self, iter, ts, Pt, error, λ):
        """src/calibrate.py:269-274 (A and k of the current parameters)"""
        nk = len(self._distortionModel.getDistortionSymbols())
        α, β, γ, uc, vc = Pt[:5]
        At = np.array([[α, γ, uc], [0, β, vc], [0, 0, 1]])
        print(f"\niter {iter}: ({time.time() - ts:0.3f}s), error={error:0.3f}, λ={λ:e}")
        print(f"A:\n{At}")
        print(f"k:\n{Pt[5:5 + nk]}")


def getSensorPoints(allDetections):
    """(MN,2) stack of every view's sensor points (src/calibrate.py:277-282)"""
    if len(allDetections) == 0:
        return np.empty((0, 2))
    return np.vstack([np.asarray(s, dtype=np.float64).reshape(-1, 2) for s, m in allDetections])
