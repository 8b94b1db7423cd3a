"""ctypes binding of include/calib_lm.h (the gfx950 shared library).

There is no CPU fallback: if the library is missing or no HIP device is
visible, every compute entry point raises.
"""
import ctypes
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CALIB_LM_LIBRARY") or os.path.join(_HERE, "lib", "libcalib_lm.so")

MODEL_RADTAN, MODEL_FISHEYE = 0, 1
DTYPE_F64, DTYPE_F32 = 0, 1
E_INVALID, E_HIP, E_SINGULAR, E_STATE = -1, -2, -3, -4
TRACE_HEADER = 5
LM_FUSED, LM_TWO_KERNEL = 0, 1

_c_double_p = ctypes.POINTER(ctypes.c_double)
_c_int64_p = ctypes.POINTER(ctypes.c_int64)
_c_int_p = ctypes.POINTER(ctypes.c_int)
_h = ctypes.c_void_p

# name -> (restype, argtypes); must list every symbol include/calib_lm.h declares
SIGNATURES = {
    "calib_version": (ctypes.c_int, []),
    "calib_last_error": (ctypes.c_char_p, []),
    "calib_device_count": (ctypes.c_int, [_c_int_p]),
    "calib_create": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(_h)]),
    "calib_destroy": (ctypes.c_int, [_h]),
    "calib_set_stream": (ctypes.c_int, [_h, ctypes.c_void_p, ctypes.c_int]),
    "calib_set_problem": (ctypes.c_int, [_h, ctypes.c_int64, _c_int64_p, _c_double_p, _c_double_p]),
    "calib_set_problem_views": (ctypes.c_int, [_h, ctypes.c_int64, _c_int64_p, ctypes.c_void_p, ctypes.c_void_p]),
    "calib_synchronize": (ctypes.c_int, [_h]),
    "calib_set_lm_mode": (ctypes.c_int, [_h, ctypes.c_int]),
    "calib_fused_form": (ctypes.c_int, [_h, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]),
    "calib_num_shared": (ctypes.c_int, [_h, _c_int_p]),
    "calib_num_params": (ctypes.c_int, [_h, _c_int64_p]),
    "calib_eval": (ctypes.c_int, [_h, _c_double_p, _c_double_p, _c_double_p, _c_double_p, _c_double_p]),
    "calib_normal_eq": (ctypes.c_int, [_h, _c_double_p, _c_double_p, _c_double_p, _c_double_p, _c_double_p]),
    "calib_lm_step_delta": (ctypes.c_int, [_h, _c_double_p, ctypes.c_double, _c_double_p]),
    "calib_refine": (ctypes.c_int, [_h, _c_double_p, ctypes.c_int, ctypes.c_double, ctypes.c_double,
                                    ctypes.c_double, ctypes.c_double, _c_double_p, _c_int_p, _c_double_p]),
    "calib_compose_params": (ctypes.c_int, [ctypes.c_int, ctypes.c_int64, _c_double_p, _c_double_p, _c_double_p,
                                            _c_double_p, ctypes.c_int]),
    "calib_decompose_params": (ctypes.c_int, [ctypes.c_int, ctypes.c_int64, _c_double_p, _c_double_p, _c_double_p,
                                              _c_double_p, ctypes.c_int]),
    "calib_refine_awk": (ctypes.c_int, [_h, _c_double_p, _c_double_p, _c_double_p, ctypes.c_int, ctypes.c_double,
                                        ctypes.c_double, ctypes.c_double, ctypes.c_double, _c_double_p, _c_int_p,
                                        _c_double_p]),
    "calib_lm_begin": (ctypes.c_int, [_h, _c_double_p, ctypes.c_int, ctypes.c_double, ctypes.c_double,
                                      ctypes.c_double, ctypes.c_double]),
    "calib_lm_reduce_size": (ctypes.c_int, [_h, _c_int64_p]),
    "calib_lm_bind_reduce_buffer": (ctypes.c_int, [_h, ctypes.c_void_p]),
    "calib_lm_local": (ctypes.c_int, [_h]),
    "calib_lm_update": (ctypes.c_int, [_h]),
    "calib_lm_run": (ctypes.c_int, [_h, ctypes.c_int, ctypes.c_int]),
    "calib_lm_run_sharded": (ctypes.c_int, [_h, ctypes.c_int, ctypes.c_int]),
    "calib_lm_done": (ctypes.c_int, [_h, _c_int_p]),
    "calib_lm_peek_trace": (ctypes.c_int, [_h, ctypes.c_int, _c_double_p, _c_int_p]),
    "calib_lm_end": (ctypes.c_int, [_h, _c_double_p, _c_double_p, _c_int_p, _c_double_p]),
    "calib_distort_points": (ctypes.c_int, [ctypes.c_int, ctypes.c_int64, _c_double_p, _c_double_p, _c_double_p]),
    "calib_project_with_distortion": (ctypes.c_int, [ctypes.c_int, ctypes.c_int64, _c_double_p, _c_double_p,
                                                     _c_double_p, _c_double_p]),
    "calib_refine_homographies": (ctypes.c_int, [ctypes.c_int64, _c_int64_p, _c_double_p, _c_double_p, _c_double_p,
                                                 ctypes.c_int, ctypes.c_int]),
    "calib_homography_jacobian": (ctypes.c_int, [ctypes.c_int64, _c_double_p, _c_double_p, _c_double_p, ctypes.c_int]),
    "calib_estimate_homographies": (ctypes.c_int, [ctypes.c_int64, _c_int64_p, _c_double_p, _c_double_p, _c_double_p,
                                                   ctypes.c_int, ctypes.c_int]),
    "calib_compute_extrinsics": (ctypes.c_int, [ctypes.c_int64, _c_double_p, _c_double_p, _c_double_p, ctypes.c_int]),
    "calib_distortion_normal_equations": (ctypes.c_int, [ctypes.c_int, ctypes.c_int64, _c_int64_p, _c_double_p,
                                                         _c_double_p, _c_double_p, _c_double_p, _c_double_p,
                                                         _c_double_p, ctypes.c_int]),
    "calib_rccl_load": (ctypes.c_int, [ctypes.c_char_p]),
    "calib_rccl_unique_id": (ctypes.c_int, [ctypes.c_void_p]),
    "calib_rccl_init": (ctypes.c_int, [_h, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "calib_rccl_init_deadline": (ctypes.c_int, [_h, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_double]),
    "calib_peer_prepare": (ctypes.c_int, [_h, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "calib_peer_connect": (ctypes.c_int, [_h, ctypes.c_void_p, ctypes.c_double]),
    "calib_peer_selftest": (ctypes.c_int, [_h, ctypes.c_int, ctypes.c_double]),
    "calib_peer_shutdown": (ctypes.c_int, [_h]),
    "calib_rccl_selftest": (ctypes.c_int, [_h, ctypes.c_double]),
    "calib_rccl_shutdown": (ctypes.c_int, [_h]),
    "calib_lm_allreduce": (ctypes.c_int, [_h]),
    "calib_profile_enable": (ctypes.c_int, [_h, ctypes.c_int]),
    "calib_profile_read": (ctypes.c_int, [_h, ctypes.c_int, _c_double_p, _c_int64_p]),
}

_lib = None


class CalibNativeError(RuntimeError):
    pass


def loadLibrary():
    """Load libcalib_lm.so (built by __graft_entry__.build() / csrc/Makefile)."""
    global _lib
    if _lib is not None:
        return _lib
    # Multi-process GPU work (RCCL, the peer exchange's hipIpc* mappings) needs the dmabuf form of HIP IPC on
    # hosts whose driver supports no other: without it hipIpcGetMemHandle fails with "invalid argument". The
    # variable is read when the HIP runtime initialises, so it is set here, before the runtime is loaded (a
    # value the caller exported wins).
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if "torch" not in sys.modules:
        # PyTorch-ROCm ships its own HIP runtime (torch/lib/libamdhip64.so). When this library is
        # loaded first it binds /opt/rocm's copy and a later `import torch` finds no devices
        # ("No HIP GPUs are available"); with torch first, both share torch's runtime. torch is
        # only plumbing here (streams, torch.distributed), so a missing torch is not an error.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    if not os.path.exists(LIB_PATH):
        raise CalibNativeError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def lastError():
    msg = loadLibrary().calib_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(rc):
    """Map a status code to the exception type the reference would raise."""
    if rc == 0:
        return
    msg = lastError()
    if rc == E_SINGULAR:
        raise np.linalg.LinAlgError(msg or "Singular matrix")      # src/calibrate.py:152 (np.linalg.inv)
    if rc == E_INVALID:
        raise ValueError(msg)                                       # src/mathutils.py:102-105
    raise CalibNativeError(f"calib_lm error {rc}: {msg}")


def deviceCount():
    n = ctypes.c_int(0)
    rc = loadLibrary().calib_device_count(ctypes.byref(n))
    if rc != 0:
        return 0
    return n.value


def requireDevice():
    if deviceCount() < 1:
        raise CalibNativeError("no HIP device visible: the LM engine runs on MI355X only "
                               f"(no CPU fallback). {lastError()}")


def dptr(a):
    """double* of a C-contiguous float64 array (None -> NULL)."""
    if a is None:
        return None
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_c_double_p)


def i64ptr(a):
    assert a.dtype == np.int64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_c_int64_p)
