"""User-facing entry point: one call from detections to (sse, A, W, k).

Same name, arguments and return value as the reference's facade (src/main.py:11-36), so callers
only change their import.
"""
from . import calibrate
from . import distortion

_MODELS = {
    "radtan": distortion.RadialTangentialModel,
    "fisheye": distortion.FisheyeModel,
}


def calibrateCamera(allDetections: list, distortionType: str, maxIters, **engineOptions) -> tuple:
    """allDetections: per view a (sensorPoints (N,2), modelPoints (N,3)) pair; distortionType one of
    "radtan", "fisheye"; engineOptions: dtype="f64"|"f32", device=<HIP device> (extras).

    Host closed-form initialisation, then the Levenberg-Marquardt refinement on the GPU.
    Returns the final sum of squared errors, the intrinsic matrix (3,3), the list of world-to-camera
    transforms (4,4) and the distortion coefficients."""
    try:
        modelClass = _MODELS[distortionType]
    except KeyError:
        raise ValueError(f"Distortion type: {distortionType} unknown") from None
    return calibrate.Calibrator(modelClass(), **engineOptions).calibrate(allDetections, maxIters)
