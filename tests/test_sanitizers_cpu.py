"""Host-side sanitizers: the C restatement of the oracle and the host side of the C99 ABI
example run under AddressSanitizer + UndefinedBehaviorSanitizer in the build container. Never on the GPU box:
GPU sanitizers are not available on this pool, and these tests need no device."""
import os
import struct
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, loadGolden


def _runtime(name):
    p = subprocess.run(["gcc", f"-print-file-name={name}"], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


ASAN, UBSAN = _runtime("libasan.so"), _runtime("libubsan.so")
needsSanitizers = pytest.mark.skipif(ASAN is None or UBSAN is None, reason="gcc's libasan / libubsan are not installed")
SAN_ENV = dict(ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")


def _clean(text):
    return not any(w in text for w in ("AddressSanitizer", "runtime error:", "UndefinedBehaviorSanitizer"))


@needsSanitizers
def test_c_oracle_golden_vectors_under_asan_ubsan():
    """tests/test_oracle_golden.py's C-oracle cases (config 1 radtan / fisheye vs the reference's vectors, the ragged,
    realistic and 200-view goldens) against oracle/_asan/libcalib_oracle.so = calib_oracle.c compiled with
    -fsanitize=address,undefined: every heap / stack access and every arithmetic operation of the checker is checked."""
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], check=True, capture_output=True)
    lib = os.path.join(ROOT, "oracle", "_asan", "libcalib_oracle.so")
    env = dict(os.environ, LD_PRELOAD=f"{ASAN}:{UBSAN}", CALIB_ORACLE_LIBRARY=lib, OMP_NUM_THREADS="4", **SAN_ENV)
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_oracle_golden.py"), "-q", "-x",
                          "-k", "c_oracle", "-p", "no:cacheprovider"], capture_output=True, text=True, env=env, cwd=ROOT, timeout=900)
    text = out.stdout + out.stderr
    assert out.returncode == 0 and " passed" in text, text[-3000:]
    assert _clean(text), text[-3000:]


@needsSanitizers
def test_c99_abi_example_host_side_under_asan_ubsan(tmp_path):
    """tests/c_abi/refine_example.c compiled with -fsanitize=address,undefined against the header and the real
    library, run WITHOUT a GPU: it parses the whole problem file (all of its host-side buffer handling) and must
    stop at its first device call (calib_device_count) with the library's own error, not with a sanitizer report. (With a GPU the same program is
    tests/test_gpu_parity.py::test_c_program_against_the_header.)"""
    import camera_calibration_amd as cca
    if cca._native.deviceCount() > 0:
        pytest.skip("a GPU is visible: the device run of this program is test_c_program_against_the_header")
    g = loadGolden("g3_unittest15.npz")
    offs, M = g["viewOffsets"].astype(np.int64), 15
    prob, res, exe = tmp_path / "problem.bin", tmp_path / "result.bin", tmp_path / "refine_example_asan"
    with open(prob, "wb") as f:
        f.write(struct.pack("<qq", 0, M))
        for a in (offs, g["sensorPoints"], g["modelPoints"], g["A0"], g["W0"], g["k0"]):
            f.write(np.ascontiguousarray(a, dtype=a.dtype if a.dtype == np.int64 else np.float64).tobytes())
    libdir = os.path.join(ROOT, "camera-calibration_amd", "lib")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-O1", "-g", "-fsanitize=address,undefined",
                    "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "c_abi", "refine_example.c"), "-o", str(exe),
                    "-L", libdir, "-lcalib_lm", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    out = subprocess.run([str(exe), str(prob), str(res)], capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, **SAN_ENV))
    text = out.stdout + out.stderr
    assert _clean(text), text[-3000:]
    assert out.returncode == 2 and "calib_device_count" in text and "no ROCm-capable device" in text, text[-2000:]
