"""camera-calibration_amd: MI355X-native Levenberg-Marquardt refinement for Zhang calibration.

Drop-in for the nonlinear stage of pvphan/camera-calibration (src/calibrate.py,
src/jacobian.py, src/distortion.py). Import as ``camera_calibration_amd`` (the
repo-root shim maps the hyphenated directory name onto that module name).
"""
from . import (_native, calibrate, dataset, distortion, engine, jacobian, linearcalibrate, main,  # noqa: F401
               mathutils, synthetic)
from .calibrate import Calibrator, getSensorPoints  # noqa: F401
from .distortion import FisheyeModel, RadialTangentialModel  # noqa: F401
from .engine import RefineEngine  # noqa: F401
from .jacobian import HomographyJacobian, ProjectionJacobian, createJacRadTan  # noqa: F401
from .main import calibrateCamera  # noqa: F401

__version__ = "0.1.0"
