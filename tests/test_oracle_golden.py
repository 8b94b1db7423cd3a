"""The CPU oracle (oracle/calib_oracle.py) against vectors produced by RUNNING THE
REFERENCE (tools/oracle/make_golden.py). This is what pins the oracle."""
import numpy as np
import pytest

from conftest import loadGolden
from oracle import calib_oracle as orc

MODELS = [("radtan", orc.RADTAN), ("fisheye", orc.FISHEYE)]


def colRel(a, b):
    scale = np.abs(b).max(axis=0)
    scale[scale == 0] = 1.0
    return (np.abs(a - b).max(axis=0) / scale).max()


def test_g0_rotations_known_answers():
    g = loadGolden("g0_mathutils.npz")
    assert np.abs(orc.eulerToR(g["angles"]) - g["R"]).max() < 1e-15
    assert np.abs(orc.rToEuler(g["R"]) - g["eulerBack"]).max() < 1e-12


@pytest.mark.parametrize("name,model", MODELS)
def test_g1_projection_and_jacobian_blocks(name, model):
    g = loadGolden("g1_blocks.npz")
    intr = g[f"{name}_intr"]
    L = intr.shape[0]
    for j in range(g[f"{name}_ext"].shape[0]):
        mp = g[f"{name}_modelPoints"][j]
        P = np.concatenate((intr, g[f"{name}_ext"][j]))
        offs = [0, mp.shape[0]]
        y = orc.projectAllPoints(model, P, offs, mp)
        Jc = orc.jacobianCompact(model, P, offs, mp)
        assert np.abs(y - g[f"{name}_y"][j]).max() < 1e-12 * 1500
        assert colRel(Jc[:, :, :L].reshape(-1, L), g[f"{name}_JI"][j]) < 1e-13
        assert colRel(Jc[:, :, L:].reshape(-1, 6), g[f"{name}_JE"][j]) < 1e-13


def test_g1_zero_parameter_case():
    # tests/test_jacobian.py:19-24: gamma = p1 = p2 = k3 = 0, rho_y = rho_z = 0
    g = loadGolden("g1_blocks.npz")
    P = np.concatenate((g["zero_intr"], g["zero_ext"]))
    Jc = orc.jacobianCompact(orc.RADTAN, P, [0, 2], g["zero_modelPoints"])
    assert not np.isnan(Jc).any()
    assert np.abs(Jc[:, :, :10].reshape(-1, 10) - g["zero_JI"]).max() < 1e-12
    assert np.abs(Jc[:, :, 10:].reshape(-1, 6) - g["zero_JE"]).max() < 1e-11
    assert np.abs(orc.projectAllPoints(orc.RADTAN, P, [0, 2], g["zero_modelPoints"]) - g["zero_y"]).max() < 1e-12


@pytest.mark.parametrize("name,model", MODELS)
def test_g2_dense_step_and_lm_trace(name, model):
    g = loadGolden(f"g2_config1_{name}.npz")
    offs, s, m, P0 = g["viewOffsets"], g["sensorPoints"], g["modelPoints"], g["P0"]
    assert np.array_equal(orc.composeParameterVector(g["A0"], g["W0"], g["k0"]), P0)
    J = orc.jacobianDense(model, P0, offs, m)
    assert J.shape == g["J"].shape
    assert np.abs(J - g["J"]).max() / np.abs(g["J"]).max() < 1e-14
    assert J[0, orc.numShared(model) + 6] == 0          # tests/test_jacobian.py:70 block sparsity
    assert np.abs(orc.projectAllPoints(model, P0, offs, m) - g["y0"]).max() < 1e-11
    assert abs(orc.reprojectionError(model, P0, offs, s, m) - g["err0"]) < 1e-9 * g["err0"]
    JTJ = J.T @ J
    assert np.abs(JTJ - g["JTJ"]).max() / np.abs(g["JTJ"]).max() < 1e-13
    # Schur step == the reference's dense inv() step
    d = orc.lmStepSchur(model, P0, offs, s, m, 1e-3)
    assert np.linalg.norm(d - g["delta"]) / np.linalg.norm(g["delta"]) < 1e-8
    # whole loop, reference-exact dense form: same iteration count, same lambda sequence
    sse, P, trace = orc.refineDense(model, P0, offs, s, m, int(g["maxIters"]))
    ref = g["traceIterErrLam"]
    assert trace.shape[0] == ref.shape[0]
    assert np.array_equal(trace[:, 3], ref[:, 2])
    early = slice(0, 5)
    assert np.allclose(np.minimum(trace[early, 1], trace[early, 2]), ref[early, 1], rtol=1e-6)
    assert sse < 1e-9 and g["sseFinal"] < 1e-9
    A, W, k = orc.decomposeParameterVector(P, model)
    assert np.abs(A - g["Afinal"]).max() < 1e-9
    assert np.abs(k - g["kfinal"]).max() < 1e-9
    assert np.abs(W - g["Wfinal"]).max() < 1e-9
    # Schur-form loop lands on the same answer
    sse2, P2, trace2 = orc.refineSchur(model, P0, offs, s, m, int(g["maxIters"]))
    A2, W2, k2 = orc.decomposeParameterVector(P2, model)
    assert sse2 < 1e-9 and np.abs(A2 - g["Afinal"]).max() < 1e-9 and np.abs(k2 - g["kfinal"]).max() < 1e-9


def test_g3_ragged_unit_test_dataset():
    g = loadGolden("g3_unittest15.npz")
    offs, s, m, P0 = g["viewOffsets"], g["sensorPoints"], g["modelPoints"], g["P0"]
    assert len(set(np.diff(offs).tolist())) > 1          # ragged
    # tests/test_calibrate.py:63-78 compose/decompose round trip, :123-133 zero error at truth
    A, W, k = orc.decomposeParameterVector(g["Ptrue"], orc.RADTAN)
    assert np.abs(W - g["WfromPtrue"]).max() < 1e-12
    assert np.abs(W - g["Wtrue"]).max() < 1e-9
    assert orc.reprojectionError(orc.RADTAN, g["Ptrue"], offs, s, m) < 1e-7
    d = orc.lmStepSchur(orc.RADTAN, P0, offs, s, m, 1e-3)
    assert np.linalg.norm(d - g["delta"]) / np.linalg.norm(g["delta"]) < 1e-8
    Jc = orc.jacobianCompact(orc.RADTAN, P0, offs, m)
    r = s - orc.projectAllPoints(orc.RADTAN, P0, offs, m)
    B, E, V, gg = orc.normalBlocks(orc.RADTAN, Jc, r, offs)
    JTJ = g["JTJ"]
    assert np.abs(B - JTJ[:10, :10]).max() / np.abs(JTJ[:10, :10]).max() < 1e-12
    for i in (0, 7, 14):
        c = 10 + 6 * i
        assert np.abs(V[i] - JTJ[c:c + 6, c:c + 6]).max() / np.abs(V[i]).max() < 1e-12
        assert np.abs(E[i] - JTJ[:10, c:c + 6]).max() / np.abs(E[i]).max() < 1e-12
    assert np.abs(JTJ[10:16, 16:22]).max() == 0          # exactly block-arrow
    assert np.abs(gg - g["JTr"]).max() / np.abs(g["JTr"]).max() < 1e-12
    sse, P, trace = orc.refineSchur(orc.RADTAN, P0, offs, s, m, 100)
    assert trace.shape[0] == g["traceIterErrLam"].shape[0]
    A, W, k = orc.decomposeParameterVector(P, orc.RADTAN)
    assert np.abs(A - g["Afinal"]).max() < 1e-9 and np.abs(k - g["kfinal"]).max() < 1e-9


def test_g4_realistic_final_answer():
    # tests/itest_main.py:12-29: A and k to 1e-9 absolute
    g = loadGolden("g4_realistic.npz")
    offs, s, m, P0 = g["viewOffsets"], g["sensorPoints"], g["modelPoints"], g["P0"]
    sse, P, trace = orc.refineSchur(orc.RADTAN, P0, offs, s, m, 100)
    A, W, k = orc.decomposeParameterVector(P, orc.RADTAN)
    assert sse < 1e-7
    assert np.abs(A - g["Atrue"]).max() < 1e-9 and np.abs(k - g["ktrue"]).max() < 1e-9
    assert np.abs(A - g["Afinal"]).max() < 1e-9 and np.abs(k - g["kfinal"]).max() < 1e-9


@pytest.mark.parametrize("tag", ["g5_ragged200.npz", "g5_ragged1000.npz"])
def test_g5_ragged_schur_vs_dense_reference_step(tag):
    """one dense reference step (dense J, J.T @ J, inv) on 200 and on 1000 ragged views -- config 2's scale,
    where the reference needs ~6 minutes and 10 GB for it -- against the oracle's Schur form"""
    g = loadGolden(tag)
    offs, s, m, P0 = g["viewOffsets"], g["sensorPoints"], g["modelPoints"], g["P0"]
    d = orc.lmStepSchur(orc.RADTAN, P0, offs, s, m, float(g["lam"]))
    assert np.linalg.norm(d - g["delta"]) / np.linalg.norm(g["delta"]) < 1e-8
    assert abs(orc.reprojectionError(orc.RADTAN, P0 + d, offs, s, m) - g["err1"]) < 1e-6 * g["err1"]
    Jc = orc.jacobianCompact(orc.RADTAN, P0, offs, m)
    n0 = int(offs[1])
    assert np.abs(Jc[:n0, :, :10].reshape(-1, 10) - g["Jview0"][:, :10]).max() < 1e-9
    assert np.abs(Jc[:n0, :, 10:].reshape(-1, 6) - g["Jview0"][:, 10:16]).max() < 1e-9


def test_g6_generator_poses():
    g = loadGolden("g6_generator.npz")
    for tag, board in (("c2", (9, 6, 0.05)), ("c3", (20, 10, 0.03)), ("c5", (11, 8, 0.04))):
        corners = orc.checkerboardCorners(*board)
        assert np.array_equal(corners, g[f"{tag}_corners"])
        W = orc.syntheticBoardPoses(corners, range(12))
        assert np.abs(W - g[f"{tag}_W"]).max() < 1e-14


@pytest.fixture(scope="module")
def corc():
    """oracle/calib_oracle.c (gcc + OpenMP), built on demand"""
    import subprocess
    from conftest import ROOT
    import os
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, capture_output=True)
    from oracle import c_oracle
    return c_oracle


@pytest.mark.parametrize("name,model", MODELS)
def test_c_oracle_matches_reference_vectors(corc, name, model):
    g = loadGolden(f"g2_config1_{name}.npz")
    offs, s, m, P0 = g["viewOffsets"], g["sensorPoints"], g["modelPoints"], g["P0"]
    L = orc.numShared(model)
    ev = corc.evaluate(model, P0, offs, s, m, wantJ=True)
    assert np.abs(ev["y"] - g["y0"]).max() < 1e-11
    assert abs(ev["sse"] - g["err0"]) < 1e-11 * g["err0"]
    Jd = g["J"]                                           # the reference's dense J
    n = ev["Jc"].shape[0]
    assert np.abs(ev["Jc"][:, :, :L].reshape(2 * n, L) - Jd[:, :L]).max() / np.abs(Jd).max() < 1e-14
    d = corc.step(model, P0, offs, s, m, 1e-3)
    assert np.linalg.norm(d - g["delta"]) / np.linalg.norm(g["delta"]) < 1e-8
    sse, P, trace = corc.refine(model, P0, offs, s, m, int(g["maxIters"]))
    ref = g["traceIterErrLam"]
    assert trace.shape[0] == ref.shape[0] and np.array_equal(trace[:, 3], ref[:, 2])
    assert np.abs(P[:L] - g["Pfinal"][:L]).max() < 1e-9


def test_c_oracle_ragged_and_realistic(corc):
    for tag in ("g3_unittest15.npz", "g4_realistic.npz", "g5_ragged200.npz"):
        g = loadGolden(tag)
        offs, s, m, P0 = g["viewOffsets"], g["sensorPoints"], g["modelPoints"], g["P0"]
        d = corc.step(orc.RADTAN, P0, offs, s, m, 1e-3)
        assert np.linalg.norm(d - g["delta"]) / np.linalg.norm(g["delta"]) < 1e-8
        sse, P, trace = corc.refine(orc.RADTAN, P0, offs, s, m, 100)
        target = g["Pfinal"][:10] if "Pfinal" in g.files else g["Ptrue"][:10]
        assert sse < 1e-7 and np.abs(P[:10] - target).max() < 1e-9
