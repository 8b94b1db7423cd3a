"""N > 1 with REAL engines: fresh child processes (spawn start method; every child initialises the GPU
itself), one rank each, all on GPU 0, through camera-calibration_amd/distributed.refineDistributed with its
default factories -- RefineEngine shards and, for the one exchange per LM round, either
torch.distributed.all_reduce on the bound reduce buffer (carried by gloo: a single GPU cannot host several
RCCL ranks) or the peer exchange inside the reduce kernel (calib_peer_*: IPC-mapped slot memory; here the
"peers" are processes on the same card). The nccl binding itself is covered at world size 1, on both the
torch and the in-library path. The parent plus at most 4 children use the card at once (the GPU boxes allow 6)."""
import os
import socket

import numpy as np
import pytest

from conftest import loadGolden

pytestmark = pytest.mark.gpu


def _freePort():
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


def _problem(case):
    if case == "g3":
        g = loadGolden("g3_unittest15.npz")
        return "radtan", g["P0"], g["viewOffsets"], g["sensorPoints"], g["modelPoints"], 10
    g = loadGolden("g2_config1_fisheye.npz")          # "fisheye3": 3 views of config 1 -> a rank without views at world 4
    offs, L, nv = g["viewOffsets"], 9, 3
    n = int(offs[nv])
    return ("fisheye", np.concatenate((g["P0"][:L], g["P0"][L:L + 6 * nv])), offs[:nv + 1], g["sensorPoints"][:n],
            g["modelPoints"][:n], L)


def _worker(rank, world, port, backend, case, allreduce, outDir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOCAL_RANK="0", RANK=str(rank),
                      WORLD_SIZE=str(world), CALIB_ALLREDUCE=allreduce, HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch
    import torch.distributed as dist
    from camera_calibration_amd import distributed
    if backend == "nccl":
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        model, P0, offs, s, m, L = _problem(case)
        sse, P, iters, trace = distributed.refineDistributed(model, P0, offs, s, m, 60)
        np.savez(os.path.join(outDir, f"r{rank}.npz"), sse=sse, P=P, iters=iters, trace=trace,
                 kind=distributed.refineDistributed.lastAllReduce)
    finally:
        dist.destroy_process_group()


def _run(tmp_path, world, backend, case, allreduce="torch"):
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(world, _freePort(), backend, case, allreduce, str(tmp_path)), nprocs=world, join=True)
    return [np.load(os.path.join(tmp_path, f"r{r}.npz")) for r in range(world)]


def _single(case):
    import camera_calibration_amd as cca
    model, P0, offs, s, m, L = _problem(case)
    eng = cca.RefineEngine(model, "f64")
    eng.setProblem(offs, s, m)
    out = eng.refine(P0, 60)
    eng.close()
    return out, L


@pytest.mark.parametrize("world,case,allreduce", [(2, "g3", "torch"), (3, "g3", "torch"), (4, "fisheye3", "torch"),
                                                  (2, "g3", "peer"), (3, "g3", "peer"), (4, "fisheye3", "peer")])
def test_refine_distributed_real_engines_on_one_gpu(tmp_path, world, case, allreduce):
    outs = _run(tmp_path, world, "gloo", case, allreduce)
    (sseR, PR, itR, trR), L = _single(case)
    for o in outs:                               # every rank returns the same global answer
        assert str(o["kind"]) == allreduce       # the requested carrier passed its self-test on every rank
        assert np.array_equal(o["P"], outs[0]["P"]) and int(o["iters"]) == int(outs[0]["iters"])
        assert np.array_equal(o["trace"], outs[0]["trace"])
    P, iters = outs[0]["P"], int(outs[0]["iters"])
    assert P.shape == PR.shape
    assert float(outs[0]["sse"]) < 1e-9 and sseR < 1e-9
    # vs the unsharded engine: the sums are formed in a different order, so compare the well-separated early
    # trace exactly (lambda) / tightly (errors) and the converged parameters
    n = min(5, iters, itR)
    assert np.array_equal(outs[0]["trace"][:n, 3], trR[:n, 3])
    assert np.allclose(outs[0]["trace"][:n, 1:3], trR[:n, 1:3], rtol=1e-9)
    assert abs(iters - itR) <= 2
    assert np.abs(P[:L] - PR[:L]).max() <= 1e-9 * max(1.0, np.abs(PR[:L]).max())
    assert np.abs(P - PR).max() <= 1e-7 * max(1.0, np.abs(PR).max())
    if case == "g3":                             # and the reference's own result
        g = loadGolden("g3_unittest15.npz")
        assert np.abs(P[:L] - g["Pfinal"][:L]).max() < 1e-9


@pytest.mark.parametrize("allreduce", ["torch", "direct"])
def test_refine_distributed_nccl_world_size_1(tmp_path, allreduce):
    """The default branch of refineDistributed over the nccl (= RCCL) backend in a fresh process: torch's
    all_reduce on the bound buffer, and the in-library ncclAllReduce (self-tested at start-up)."""
    outs = _run(tmp_path, 1, "nccl", "g3", allreduce)
    assert str(outs[0]["kind"]) == allreduce
    (sseR, PR, itR, trR), L = _single("g3")
    assert int(outs[0]["iters"]) == itR and np.array_equal(outs[0]["P"], PR) and np.array_equal(outs[0]["trace"], trR)


def _peerWorker(rank, world, port, case, outDir):
    """calib_peer_* alone: prepare / connect / self-test, a lost rank (bounded spin), shutdown."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    import camera_calibration_amd as cca
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = {"error": "", "ok": 0}
    try:
        eng = cca.RefineEngine("radtan", "f64")
        handles = [None] * world
        dist.all_gather_object(handles, eng.peerPrepare(world, rank))
        assert all(len(hd) == 64 for hd in handles)
        eng.peerConnect(handles, 5.0)
        eng.peerSelfTest(200, 20.0)
        out["ok"] = 1
        if case == "lost_rank":
            # rank 1 does not take part in a further exchange: the others stop waiting after the deadline and say so
            dist.barrier()
            if rank != 1:
                # 40 rounds are enqueued at once with a 0.5 s deadline each: only the first may spin it out, the rest
                # see the fault word and return at once (they used to wait 40 x 0.5 s)
                import time
                t0 = time.perf_counter()
                try:
                    eng.peerSelfTest(40, 0.5)
                except RuntimeError as e:
                    out["error"] = str(e)
                out["lost_seconds"] = time.perf_counter() - t0
            dist.barrier()
        eng.peerShutdown()
        eng.close()
    finally:
        np.savez(os.path.join(outDir, f"p{rank}.npz"), **out)
        dist.destroy_process_group()


@pytest.mark.parametrize("world,case", [(4, "selftest"), (2, "lost_rank")])
def test_peer_exchange_selftest_and_bounded_spin(tmp_path, world, case):
    import torch.multiprocessing as mp
    mp.spawn(_peerWorker, args=(world, _freePort(), case, str(tmp_path)), nprocs=world, join=True)
    outs = [np.load(os.path.join(tmp_path, f"p{r}.npz")) for r in range(world)]
    assert all(int(o["ok"]) == 1 for o in outs)
    if case == "lost_rank":
        assert "did not arrive" in str(outs[0]["error"]) and str(outs[1]["error"]) == ""
        assert float(outs[0]["lost_seconds"]) < 5.0, float(outs[0]["lost_seconds"])
