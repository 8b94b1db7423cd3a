"""Import shim: ``import camera_calibration_amd`` loads the package that lives in the
directory ``camera-calibration_amd/`` (a hyphen is not importable as written)."""
import importlib.util
import os
import sys

_pkg_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "camera-calibration_amd")
_spec = importlib.util.spec_from_file_location(
    "camera_calibration_amd", os.path.join(_pkg_dir, "__init__.py"),
    submodule_search_locations=[_pkg_dir])
_module = importlib.util.module_from_spec(_spec)
sys.modules["camera_calibration_amd"] = _module
_spec.loader.exec_module(_module)
