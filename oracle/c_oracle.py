"""ctypes binding of oracle/calib_oracle.c (test infrastructure; see that file's header)."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# CALIB_ORACLE_LIBRARY: another build of the same source (oracle/_asan/: the sanitizer build of `make -C oracle asan`)
_LIB = os.environ.get("CALIB_ORACLE_LIBRARY") or os.path.join(_HERE, "libcalib_oracle.so")
_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int64)
_lib = None


def available():
    return os.path.exists(_LIB)


def _load():
    global _lib
    if _lib is None:
        lib = ctypes.CDLL(_LIB)
        lib.oracle_eval.argtypes = [ctypes.c_int, _dp, ctypes.c_int64, _ip, _dp, _dp, _dp, _dp, _dp, _dp]
        lib.oracle_step.argtypes = [ctypes.c_int, _dp, ctypes.c_int64, _ip, _dp, _dp, ctypes.c_double, _dp]
        lib.oracle_refine.argtypes = [ctypes.c_int, _dp, ctypes.c_int64, _ip, _dp, _dp, ctypes.c_int,
                                      ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                      _dp, ctypes.POINTER(ctypes.c_int), _dp]
        _lib = lib
    return _lib


def _d(a):
    return None if a is None else a.ctypes.data_as(_dp)


def _prep(P, offs, sensor, pts):
    P = np.ascontiguousarray(P, dtype=np.float64).ravel()
    offs = np.ascontiguousarray(offs, dtype=np.int64)
    sensor = None if sensor is None else np.ascontiguousarray(sensor, dtype=np.float64)
    pts = np.ascontiguousarray(pts, dtype=np.float64)
    return P, offs, sensor, pts


def evaluate(model, P, offs, sensor, pts, wantJ=False):
    P, offs, sensor, pts = _prep(P, offs, sensor, pts)
    MN = int(offs[-1])
    L = 10 if model == 0 else 9
    y = np.empty((MN, 2))
    r = np.empty((MN, 2)) if sensor is not None else None
    Jc = np.empty((MN, 2, L + 6)) if wantJ else None
    sse = ctypes.c_double(0.0)
    _load().oracle_eval(model, _d(P), offs.shape[0] - 1, offs.ctypes.data_as(_ip), _d(sensor), _d(pts),
                        _d(y), _d(r), _d(Jc), ctypes.byref(sse))
    return {"y": y, "r": r, "Jc": Jc, "sse": sse.value}


def step(model, P, offs, sensor, pts, lam):
    P, offs, sensor, pts = _prep(P, offs, sensor, pts)
    d = np.empty_like(P)
    rc = _load().oracle_step(model, _d(P), offs.shape[0] - 1, offs.ctypes.data_as(_ip), _d(sensor), _d(pts),
                             float(lam), _d(d))
    if rc:
        raise np.linalg.LinAlgError("Singular matrix")
    return d


def refine(model, P0, offs, sensor, pts, maxIters, lamInit=1e-3, lamMin=1e-10, lamMax=1e10, errMin=1e-12):
    P, offs, sensor, pts = _prep(P0, offs, sensor, pts)
    P = P.copy()
    trace = np.zeros((maxIters, 5))
    sse = ctypes.c_double(0.0)
    iters = ctypes.c_int(0)
    rc = _load().oracle_refine(model, _d(P), offs.shape[0] - 1, offs.ctypes.data_as(_ip), _d(sensor), _d(pts),
                               int(maxIters), lamInit, lamMin, lamMax, errMin, ctypes.byref(sse),
                               ctypes.byref(iters), _d(trace))
    if rc:
        raise np.linalg.LinAlgError("Singular matrix")
    return sse.value, P, trace[:iters.value]
