"""CPU-side checks: the C-ABI library loads and exports every symbol include/calib_lm.h
declares, host packing / pose helpers agree with the reference's golden vectors, and the
product path refuses to run without a GPU (no CPU fallback)."""
import os
import re

import numpy as np
import pytest

import camera_calibration_amd as cca
from camera_calibration_amd import _native as nat
from camera_calibration_amd import engine, mathutils as mu, synthetic
from conftest import ROOT, loadGolden


def headerSymbols():
    text = open(os.path.join(ROOT, "include", "calib_lm.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(calib_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = nat.loadLibrary()
    names = headerSymbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/calib_lm.h but not exported"
    assert sorted(nat.SIGNATURES) == names, "ctypes binding and header disagree"
    assert lib.calib_version() >= 100


def test_rccl_standin_exports_what_the_library_resolves():
    """tests/fake_rccl/librccl_standin.so (test infrastructure: several ranks on ONE GPU through the in-library
    ncclAllReduce carrier) must offer exactly the entry points calib_rccl_load looks up (csrc/calib_lm.hip)."""
    import ctypes
    path = os.path.join(ROOT, "tests", "fake_rccl", "librccl_standin.so")
    assert os.path.exists(path), "make -C tests/fake_rccl (done by __graft_entry__.build())"
    lib = ctypes.CDLL(path)
    for n in ("ncclGetUniqueId", "ncclCommInitRank", "ncclAllReduce", "ncclCommDestroy", "ncclCommAbort", "ncclGetErrorString"):
        assert hasattr(lib, n), n
    src = open(os.path.join(ROOT, "camera-calibration_amd", "csrc", "calib_lm.hip")).read()
    assert sorted(set(re.findall(r'dlsym\(lib, "(nccl[A-Za-z]+)"\)', src))) == sorted(
        ["ncclGetUniqueId", "ncclCommInitRank", "ncclAllReduce", "ncclCommDestroy", "ncclCommAbort", "ncclGetErrorString"])
    ident = ctypes.create_string_buffer(128)
    assert lib.ncclGetUniqueId(ident) == 0 and ident.raw.startswith(b"/calib_rccl_standin_")
    lib.ncclGetErrorString.restype = ctypes.c_char_p
    assert b"stand-in" in lib.ncclGetErrorString(2)


def test_no_cpu_fallback_without_device():
    if nat.deviceCount() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(nat.CalibNativeError):
        cca.RefineEngine("radtan")
    with pytest.raises(nat.CalibNativeError):
        cca.RadialTangentialModel().distortPoints(np.zeros((2, 2)), (0, 0, 0, 0, 0))


def test_pose_helpers_vs_reference_known_answers():
    g = loadGolden("g0_mathutils.npz")
    assert np.abs(mu.eulerToRotationMatrices(g["angles"]) - g["R"]).max() < 1e-15
    assert np.abs(mu.rotationMatricesToEuler(g["R"]) - g["eulerBack"]).max() < 1e-12
    assert np.abs(np.array(mu.rotationMatrixToEuler(g["R"][3])) - g["eulerBack"][3]).max() < 1e-12


def test_compose_decompose_need_the_device():
    # the Euler (de)composition of the parameter vector runs behind the C-ABI (calib_compose_params /
    # calib_decompose_params); its parity test is tests/test_gpu_parity.py::test_compose_decompose_on_device
    if nat.deviceCount() > 0:
        pytest.skip("a GPU is visible")
    g = loadGolden("g3_unittest15.npz")
    cal = cca.Calibrator(cca.RadialTangentialModel())
    with pytest.raises(nat.CalibNativeError):
        cal._composeParameterVector(g["Atrue"], list(g["Wtrue"]), tuple(g["ktrue"]))
    with pytest.raises(nat.CalibNativeError):
        cal._decomposeParameterVector(g["Ptrue"])


def test_pack_detections_and_sensor_points():
    g = loadGolden("g3_unittest15.npz")
    offs = g["viewOffsets"]
    dets = [(g["sensorPoints"][a:b], g["modelPoints"][a:b]) for a, b in zip(offs[:-1], offs[1:])]
    o2, s2, m2 = engine.packDetections(dets)
    assert np.array_equal(o2, offs) and np.array_equal(s2, g["sensorPoints"]) and np.array_equal(m2, g["modelPoints"])
    assert np.array_equal(cca.getSensorPoints(dets), g["sensorPoints"])
    assert cca.getSensorPoints([]).shape == (0, 2)
    with pytest.raises(ValueError):
        engine.packDetections([(np.zeros((3, 2)), np.zeros((4, 3)))])


def test_synthetic_pose_sampler_vs_reference_dataset():
    g = loadGolden("g6_generator.npz")
    for tag in ("c2", "c3", "c5"):
        corners = synthetic.checkerboardCorners(*synthetic.CONFIGS[tag]["board"])
        assert np.array_equal(corners, g[f"{tag}_corners"])
        W = synthetic.sampleBoardPosesInCamera(corners, np.arange(12))
        assert np.abs(W - g[f"{tag}_W"]).max() < 1e-13
        # any sub-range gives the same poses (global view index)
        W2 = synthetic.sampleBoardPosesInCamera(corners, np.arange(5, 9))
        assert np.array_equal(W2, W[5:9])


@pytest.mark.parametrize("tag,name", [("g2_config1_radtan.npz", "radtan"), ("g2_config1_fisheye.npz", "fisheye"),
                                      ("g3_unittest15.npz", "radtan"), ("g4_realistic.npz", "radtan")])
def test_linear_initialisation_vs_reference(tag, name):
    """Host closed-form stage (batched numpy) against what the reference's
    estimateCalibrationParameters produced on the same detections (src/calibrate.py:41-58)."""
    from camera_calibration_amd import linearcalibrate as lc
    g = loadGolden(tag)
    offs = g["viewOffsets"]
    dets = [(g["sensorPoints"][a:b], g["modelPoints"][a:b]) for a, b in zip(offs[:-1], offs[1:])]
    model = cca.RadialTangentialModel() if name == "radtan" else cca.FisheyeModel()
    A, W, k = lc.estimateCalibrationParameters(model, dets)
    assert A.shape == (3, 3) and len(W) == len(dets) and len(k) == len(model.getDistortionSymbols())
    assert np.abs(A - g["A0"]).max() < 1e-5           # tests/test_linearcalibrate.py holds A to 1e-6 relative
    assert np.abs(np.array(W) - g["W0"]).max() < 1e-6
    assert np.abs(np.array(k) - g["k0"]).max() < 1e-5
    # homographies reproject the detections (tests/test_linearcalibrate.py:55-75, 1e-3 bar is on H itself)
    Hs = lc.refineHomographies(lc.estimateHomographies(dets), dets)
    for H, (s, m) in zip(Hs[:3], dets[:3]):
        p = np.column_stack((m[:, :2], np.ones(m.shape[0]))) @ H.T
        assert np.abs(p[:, :2] / p[:, 2:3] - s).max() < 30.0      # lens distortion is not in H
    assert abs(Hs[0][2, 2] - 1.0) < 1e-15


def test_homography_on_the_references_stored_detections():
    """The reference's one stored, non-synthetic detection set (tests/test_linearcalibrate.py:72-77 with the 57 corners of
    :266-386: real-image coordinates, missing corners) and its known-answer case (:55-70, Hexpected to 1e-3), golden g10
    = what the reference's estimateHomography / _refineHomographies return on them; here the batched host stage."""
    from camera_calibration_amd import linearcalibrate as lc
    g = loadGolden("g10_real_detections.npz")
    dets = [(g["ex_x"], g["ex_X"]), (g["ka_x"], g["ka_X"])]
    Hex, Hka = lc.estimateHomographies(dets)
    assert np.abs(Hex - g["ex_H"]).max() <= 1e-8 * np.abs(g["ex_H"]).max()
    assert np.allclose(Hka, g["ka_Hexpected"], atol=1e-3)                  # the reference's own assertion
    assert np.abs(Hka - g["ka_H"]).max() <= 1e-8 * np.abs(g["ka_H"]).max()
    Hr = lc.refineHomographies([Hex, Hka], dets)
    assert np.abs(Hr[0] - g["ex_Href"]).max() <= 1e-5 * np.abs(g["ex_Href"]).max()
    for H, Hg, (x, X) in ((Hex, g["ex_H"], dets[0]), (Hr[0], g["ex_Href"], dets[0])):
        p = np.column_stack((X[:, :2], np.ones(X.shape[0])))
        mine, theirs = p @ H.T, p @ Hg.T
        assert np.abs(mine[:, :2] / mine[:, 2:3] - theirs[:, :2] / theirs[:, 2:3]).max() < 1e-6      # px


def test_pack_detections_fast_path_and_fallback():
    """engine.packDetections = getSensorPoints' vstack done once (src/calibrate.py:277-282): float64 (N,2)/(N,3) arrays take
    one pass and one concatenate; lists / other dtypes are converted view by view; shape errors are ValueErrors that
    name the view (mathutils.validateShape's exception type, src/mathutils.py:102-105)."""
    rng = np.random.default_rng(3)
    ns = [5, 1, 17, 4]
    dets = [(rng.random((n, 2)), rng.random((n, 3))) for n in ns]
    offs, s, m = engine.packDetections(dets)
    assert offs.dtype == np.int64 and list(offs) == [0, 5, 6, 23, 27]
    assert np.array_equal(s, np.vstack([d[0] for d in dets])) and np.array_equal(m, np.vstack([d[1] for d in dets]))
    assert s.flags["C_CONTIGUOUS"] and m.flags["C_CONTIGUOUS"] and s.dtype == np.float64
    assert np.array_equal(cca.getSensorPoints(dets), s)
    mixed = [(dets[0][0].tolist(), dets[0][1].astype(np.float32)), dets[1], (dets[2][0][::1], dets[2][1]), dets[3]]
    o2, s2, m2 = engine.packDetections(mixed)
    assert np.array_equal(o2, offs) and np.array_equal(s2, s) and np.allclose(m2, m, atol=1e-7)
    views = np.vstack([d[0] for d in dets])                       # slices of one array (what bench.py hands over)
    o3, s3, _ = engine.packDetections([(views[a:b], dets[i][1]) for i, (a, b) in enumerate(zip(offs[:-1], offs[1:]))])
    assert np.array_equal(s3, s) and not np.shares_memory(s3, views)
    with pytest.raises(ValueError, match="view 1"):
        engine.packDetections([dets[0], (np.zeros((3, 2)), np.zeros((4, 3)))])
    with pytest.raises(ValueError, match="view 0"):
        engine.packDetections([(np.zeros((3, 3)), np.zeros((3, 3)))])
    o0, s0, m0 = engine.packDetections([])
    assert list(o0) == [0] and s0.shape == (0, 2) and m0.shape == (0, 3)
    om, mm = engine.packModelPoints([d[1] for d in dets])
    assert np.array_equal(om, offs) and np.array_equal(mm, m)


def test_gather_of_per_view_arrays_equals_the_stacked_matrix(tmp_path):
    """calib_set_problem_views' staged upload reads the caller's per-view arrays through HostRows::copy
    (csrc/host_rows.hpp); tests/host_cpp/host_rows_check.cpp holds it, on the host, to the np.vstack of the views
    (src/calibrate.py:277-282) for every chunking -- borders inside rows and views, empty views."""
    import subprocess
    exe = tmp_path / "host_rows_check"
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", "-o", str(exe),
                    os.path.join(ROOT, "tests", "host_cpp", "host_rows_check.cpp")], check=True, capture_output=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ok" in out.stdout, out.stdout + out.stderr


def test_view_pointers_of_a_detection_list():
    """engine.viewPointers (csrc/fastpack.c): per-view row counts and data addresses of the reference's allDetections
    list, for calib_set_problem_views to gather from -- or None (numpy stacks instead) when a view is not a C-contiguous
    float64 array of the right width; mismatched counts are the ValueError packDetections raises."""
    assert engine._fastpackModule() is not None, "lib/_fastpack*.so is built by csrc/Makefile (__graft_entry__.build())"
    rng = np.random.default_rng(4)
    ns = [5, 0, 17, 4]
    dets = [(rng.random((n, 2)), rng.random((n, 3))) for n in ns]
    offs, sa, ma = engine.viewPointers(dets)
    assert list(offs) == [0, 5, 5, 22, 26] and sa.dtype == np.uint64 and ma.dtype == np.uint64
    for i, (s_, m_) in enumerate(dets):
        assert int(sa[i]) == s_.ctypes.data and int(ma[i]) == m_.ctypes.data
    assert engine.viewPointers([(d[0], d[1]) for d in dets][:0]) is None                     # nothing to point at
    assert engine.viewPointers([(dets[0][0].tolist(), dets[0][1])]) is None                  # not an ndarray
    assert engine.viewPointers([(dets[0][0].astype(np.float32), dets[0][1])]) is None        # not float64
    assert engine.viewPointers([(dets[0][0][::2], dets[0][1][::2])]) is None                 # not contiguous
    assert engine.viewPointers([(dets[0][1], dets[0][1])]) is None                           # wrong width
    assert engine.viewPointers([dets[0][0]]) is None                                         # not a pair
    with pytest.raises(ValueError, match="view 1"):
        engine.viewPointers([dets[0], (np.zeros((3, 2)), np.zeros((4, 3)))])


def test_detections_json_roundtrip(tmp_path):
    # tests/test_dataset.py:70-90: exportDetections / createDetectionsFromPath, same JSON schema
    from camera_calibration_amd import dataset
    g = loadGolden("g3_unittest15.npz")
    offs = g["viewOffsets"]
    dets = [(g["sensorPoints"][a:b], g["modelPoints"][a:b]) for a, b in zip(offs[:-1], offs[1:])]
    path = str(tmp_path / "detections.json")
    dataset.exportDetections(dets, path)
    import json
    doc = json.load(open(path))
    assert set(doc) == {"views"} and set(doc["views"][0]) == {"sensorPoints", "modelPoints"}
    back = dataset.createDetectionsFromPath(path)
    assert len(back) == len(dets)
    for (s0, m0), (s1, m1) in zip(dets, back):
        assert np.array_equal(s0, s1) and np.array_equal(m0, m1)


def test_mathutils_helpers_vs_reference_known_answers():
    """exp / skew / unskew / stack / unstack / project / projectStandard (src/mathutils.py:59-117,149-192;
    call shapes of tests/test_mathutils.py:91-152) against outputs of the reference (golden g8)"""
    g = loadGolden("g8_surface.npz")
    for w, R, K, u in zip(g["exp_w"], g["exp_R"], g["skew"], g["unskew"]):
        assert np.array_equal(mu.skew(mu.col(w)), K)
        assert np.array_equal(mu.unskew(K), u)
        assert np.abs(mu.exp(mu.skew(mu.col(w))) - R).max() < 1e-15
    with pytest.raises(ValueError):
        mu.skew(np.zeros((3, 2)))            # validateShape, as in the reference
    with pytest.raises(ValueError):
        mu.unskew(np.zeros((3, 1)))
    assert np.array_equal(mu.stack(g["stack_A"]), g["stack_out"])
    assert np.array_equal(mu.unstack(g["stack_out"]), g["unstack_out"])
    assert np.abs(mu.project(g["project_A"], g["project_wMc"], g["project_wX"]) - g["project_u"]).max() < 1e-11
    assert np.abs(mu.projectStandard(g["projectStandard_X"]) - g["projectStandard_x"]).max() < 1e-15
    with pytest.raises(ValueError):
        mu.projectStandard(np.zeros((4, 2)))
    assert mu.radians(180.0) == np.pi and np.array_equal(mu.normalize(np.array([2.0, 4.0])), [0.5, 1.0])


def test_injected_jacobian_is_refused_before_anything_runs():
    """src/calibrate.py:144 consults self._jac every iteration; the drop-in's device loop cannot, and refuses an
    object that is not its own ProjectionJacobian instead of silently ignoring it (no device needed to say so)."""
    from unittest.mock import MagicMock
    g = loadGolden("g3_unittest15.npz")
    offs = g["viewOffsets"]
    dets = [(g["sensorPoints"][a:b], g["modelPoints"][a:b]) for a, b in zip(offs[:-1], offs[1:])]
    cal = cca.Calibrator(cca.RadialTangentialModel())
    cal._jac = MagicMock()
    with pytest.raises(TypeError, match="injected Jacobian"):
        cal.refineCalibrationParameters(g["A0"], list(g["W0"]), tuple(g["k0"]), dets, 2)
    cal._jac.compute.assert_not_called()


def test_record_emission_tables_cover_every_read_entry_once(tmp_path):
    """The stream form of the fused kernel stores a view's record straight from the block accumulators through a per-lane
    table (kernels.hpp: buildStreamOps). tests/host_cpp/stream_ops_check.cpp runs the table builders on the HOST and
    checks, for both models, that every record entry the per-view kernels read is stored by exactly one lane, nothing
    else is stored, and the source kind of each op matches the emit table."""
    import subprocess
    exe = tmp_path / "stream_ops_check"
    subprocess.run(["/opt/rocm/bin/hipcc", "-std=c++17", "-O1", "--offload-arch=gfx950", "-w", "-o", str(exe),
                    os.path.join(ROOT, "tests", "host_cpp", "stream_ops_check.cpp")], check=True, capture_output=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "C=15: 90 record entries" in out.stdout and "C=16: 96 record entries" in out.stdout   # six 16-wide rows; fisheye has no column 15
    # ... and stream_extra_item (which overflow record holds the rest of a view cut by a wave start) agrees with a
    # direct simulation of the cuts on a few thousand (groups per view, views, waves) combinations
    assert "wave cuts:" in out.stdout and out.stdout.rstrip().endswith("ok")
