"""Synthetic datasets with the reference's API (src/dataset.py, src/checkerboard.py,
src/virtualcamera.py, src/noise.py) and its detections JSON schema.

Poses are sampled on the host exactly as the reference does (legacy RNG seeded with the view index,
src/dataset.py:64-70); the projection of the board corners through the camera model runs on the
device (`RefineEngine.evaluate`), then the image crop of src/virtualcamera.py:50-54 is applied.
"""
import json

import numpy as np

from . import distortion
from . import engine
from . import synthetic


class Checkerboard:
    """src/checkerboard.py:4-22"""

    def __init__(self, numCornersWidth, numCornersHeight, spacing):
        self._cornerPositions = synthetic.checkerboardCorners(numCornersWidth, numCornersHeight, spacing)

    def getCornerPositions(self, ids=None) -> np.ndarray:
        return self._cornerPositions if ids is None else self._cornerPositions[ids]


class NoiseModel:
    """zero-mean Gaussian sensor noise from the legacy global RNG (src/noise.py:4-18)"""

    def __init__(self, standardDeviation: float):
        self._standardDeviation = standardDeviation

    def applyNoise(self, sensorPoints: np.ndarray):
        return sensorPoints + np.random.normal(0.0, self._standardDeviation, sensorPoints.shape)


class VirtualCamera:
    """src/virtualcamera.py:10-55"""

    def __init__(self, intrinsicMatrix, distortionVector, distortionModel: distortion.DistortionModel,
                 imageWidth, imageHeight, noiseModel=None):
        self._intrinsicMatrix = np.asarray(intrinsicMatrix, dtype=np.float64)
        self._distortionVector = tuple(distortionVector)
        self._distortionModel = distortionModel
        self._imageWidth = imageWidth
        self._imageHeight = imageHeight
        self._noiseModel = noiseModel

    def getIntrinsicMatrix(self):
        return self._intrinsicMatrix

    def getDistortionVector(self):
        return self._distortionVector

    def getImageWidth(self):
        return self._imageWidth

    def getImageHeight(self):
        return self._imageHeight


class Dataset:
    """src/dataset.py:17-109. `crop=False` keeps every corner of every view (benchmark shapes)."""

    def __init__(self, checkerboard: Checkerboard, virtualCamera: VirtualCamera, numViews: int,
                 viewStart=0, crop=True, device=0):
        self._checkerboard = checkerboard
        self._virtualCamera = virtualCamera
        corners = checkerboard.getCornerPositions()
        N = corners.shape[0]
        cam = virtualCamera
        W = synthetic.sampleBoardPosesInCamera(corners, np.arange(viewStart, viewStart + numViews))
        self._allBoardPosesInCamera = list(W)
        P = synthetic.composeP(cam.getIntrinsicMatrix(), W, cam.getDistortionVector())
        offs = np.arange(numViews + 1, dtype=np.int64) * N
        model = np.ascontiguousarray(np.tile(corners, (numViews, 1)))
        eng = engine.RefineEngine(cam._distortionModel.modelId, "f64", device)
        try:
            eng.setProblem(offs, None, model)
            u = eng.evaluate(P, wantY=True)["y"].reshape(numViews, N, 2)
        finally:
            eng.close()
        zc = np.einsum("vj,nj->vn", W[:, 2, :3], corners) + W[:, 2, 3:4]       # camera-frame depth
        ids = np.arange(N)
        self._allIdsDetections = []
        for i in range(numViews):
            ui = u[i]
            if cam._noiseModel is not None:
                np.random.seed(viewStart + i)          # the reference draws the noise after the pose,
                np.random.choice(N)                    # from the generator it re-seeded for this view
                for _ in range(4):
                    np.random.uniform()
                ui = cam._noiseModel.applyNoise(ui)
            if crop:
                with np.errstate(invalid="ignore"):
                    keep = ((ui[:, 0] > 0) & (ui[:, 0] < cam.getImageWidth())
                            & (ui[:, 1] > 0) & (ui[:, 1] < cam.getImageHeight()) & (zc[i] > 0))
            else:
                keep = np.ones(N, dtype=bool)
            self._allIdsDetections.append((ids[keep], ui[keep], corners[keep]))

    def getCornerDetectionsInSensorCoordinates(self):
        return [(s, m) for ids, s, m in self._allIdsDetections]

    def getAllBoardPosesInCamera(self):
        return self._allBoardPosesInCamera

    def getIntrinsicMatrix(self):
        return self._virtualCamera.getIntrinsicMatrix()

    def getDistortionVector(self):
        return self._virtualCamera.getDistortionVector()

    def getImageWidth(self):
        return self._virtualCamera.getImageWidth()

    def getImageHeight(self):
        return self._virtualCamera.getImageHeight()

    def exportDetections(self, filePath):
        exportDetections(self.getCornerDetectionsInSensorCoordinates(), filePath)


def exportDetections(allDetections, filePath):
    """{"views": [{"sensorPoints": [[u, v], ...], "modelPoints": [[X, Y, Z], ...]}, ...]}
    (src/dataset.py:97-109)"""
    views = [{"sensorPoints": np.asarray(s).tolist(), "modelPoints": np.asarray(m).tolist()}
             for s, m in allDetections]
    with open(filePath, "w") as f:
        f.write(json.dumps({"views": views}))


def createDetectionsFromPath(filePath):
    """src/dataset.py:133-141"""
    with open(filePath, "r") as f:
        detectionsDict = json.load(f)
    return [(np.array(v["sensorPoints"], dtype=np.float64).reshape(-1, 2),
             np.array(v["modelPoints"], dtype=np.float64).reshape(-1, 3)) for v in detectionsDict["views"]]


def createSyntheticDataset(A, width, height, k, distortionModel, noiseModel):
    """25 x 18 board, 30 mm spacing, 15 views (src/dataset.py:124-130)"""
    return Dataset(Checkerboard(25, 18, 0.030),
                   VirtualCamera(A, k, distortionModel, width, height, noiseModel), 15)


def createSyntheticDatasetRadTan(A, width, height, k, noiseModel):
    return createSyntheticDataset(A, width, height, k, distortion.RadialTangentialModel(), noiseModel)


def createSyntheticDatasetFisheye(A, width, height, k, noiseModel):
    return createSyntheticDataset(A, width, height, k, distortion.FisheyeModel(), noiseModel)


def createRealisticRadTanDataset():
    """src/dataset.py:144-155"""
    A = np.array([[1432.1, 0, 719.2], [0, 1432.1, 564.3], [0, 0, 1]])
    k = (-0.2674, 0.1716, 1.4287e-05, 0.000177, -0.052701)
    return createSyntheticDatasetRadTan(A, 1440, 1080, k, None)
