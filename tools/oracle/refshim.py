"""Import harness for the read-only reference at /root/reference (build container only).

This module is NEVER needed on the GPU box: it only exists so that
``tools/oracle/make_golden.py`` can run the reference's own numpy/sympy code
here and write small golden input/output vectors under ``tests/golden/``.

Two accommodations, both outside the reference's arithmetic:
  1. ``src/visualize.py:4-5`` imports cv2/imageio (debug drawing only) which
     are not installed -> empty stand-in modules in ``sys.modules``.
  2. numpy>=1.24 refuses the ragged (scalar | (N,1) array) Matrix that
     ``sympy.lambdify`` returns at ``src/jacobian.py:195-199`` (numpy 1.21,
     which the reference pins, built an object array and only warned) ->
     lambdify every matrix entry separately and return the (2,T) object
     array numpy 1.21 produced; ``structureJacobianResults`` consumes it
     unchanged.
"""
import sys
import types
import warnings

sys.dont_write_bytecode = True  # /root/reference is read-only
sys.modules.setdefault("cv2", types.ModuleType("cv2"))
sys.modules.setdefault("imageio", types.ModuleType("imageio"))
REFERENCE_SRC = "/root/reference/src"
if REFERENCE_SRC not in sys.path:
    sys.path.insert(0, REFERENCE_SRC)

import numpy as np  # noqa: E402
import sympy  # noqa: E402
from __context__ import src  # noqa: E402,F401
from src import (  # noqa: E402
    calibrate,
    checkerboard,
    dataset,
    distortion,
    jacobian,
    linearcalibrate,
    main,
    mathutils,
    noise,
    virtualcamera,
)


def _entrywise_lambdify(expression, orderedSymbols):
    expression = sympy.Matrix(expression)
    rows, cols = expression.shape
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        fs = [[sympy.lambdify(orderedSymbols, expression[r, c], "numpy")
               for c in range(cols)] for r in range(rows)]

    def f(*args):
        out = np.empty((rows, cols), dtype=object)
        for r in range(rows):
            for c in range(cols):
                out[r, c] = fs[r][c](*args)
        return out

    return f


# must be in place before any *Jacobian object is constructed
jacobian.createLambdaFunction = _entrywise_lambdify
dataset.Dataset.writeDatasetImages = lambda self, path: None

_jacCache = {}


def getCalibrator(modelName):
    """One Calibrator per model with its (slow, 5-25 s) sympy Jacobian built once."""
    if modelName not in _jacCache:
        model = {"radtan": distortion.RadialTangentialModel,
                 "fisheye": distortion.FisheyeModel}[modelName]()
        cal = calibrate.Calibrator(model)
        cal._initializeJacobian()
        _jacCache[modelName] = cal
    return _jacCache[modelName]
