"""User-facing facade (reference: src/main.py:11-36)."""
from . import calibrate
from . import distortion


def calibrateCamera(allDetections: list, distortionType: str, maxIters, **engineOptions) -> tuple:
    """Intrinsic matrix, distortion coefficients and board poses from a set of detections.

    allDetections -- list of (sensorPoints (N,2), modelPoints (N,3)), one per view
    distortionType -- "radtan" or "fisheye"
    engineOptions -- dtype="f64"|"f32", device=<HIP device index> (extras over the reference)
    -> (sse, Afinal (3,3), Wfinal list of (4,4), kFinal)
    """
    if distortionType == "radtan":
        distortionModel = distortion.RadialTangentialModel()
    elif distortionType == "fisheye":
        distortionModel = distortion.FisheyeModel()
    else:
        raise ValueError(f"Distortion type: {distortionType} unknown")
    calibrator = calibrate.Calibrator(distortionModel, **engineOptions)
    return calibrator.calibrate(allDetections, maxIters)
