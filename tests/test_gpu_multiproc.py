"""N > 1 with REAL engines: fresh child processes (spawn start method; every child initialises the GPU
itself), one rank each, all on GPU 0, through camera-calibration_amd/distributed.refineDistributed with its
default factories -- RefineEngine shards and, for the one exchange per LM round, either
torch.distributed.all_reduce on the bound reduce buffer (carried by gloo: a single GPU cannot host several
RCCL ranks) or the peer exchange inside the reduce kernel (calib_peer_*: IPC-mapped slot memory; here the
"peers" are processes on the same card), or the library's own ncclAllReduce (calib_rccl_*: the DEFAULT carrier of a
multi-GPU run) resolved from tests/fake_rccl's stand-in librccl -- the real RCCL refuses two ranks on one device, so
the stand-in (shared-memory exchange behind the same five entry points, asynchronous and stream-ordered like the
real one) is what lets that branch of calib_lm_run_sharded have more than one participant before an 8-GPU node does.
The nccl binding itself is covered at world size 1, on both the torch and the in-library path. The parent plus at
most 4 children use the card at once (the GPU boxes allow 6)."""
import os
import socket

import numpy as np
import pytest

from conftest import loadGolden

pytestmark = pytest.mark.gpu

STANDIN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fake_rccl", "librccl_standin.so")


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


def _worker(rank, world, port, backend, case, allreduce, outDir, extraEnv=None, checkEvery=8):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOCAL_RANK="0", RANK=str(rank),
                      WORLD_SIZE=str(world), CALIB_ALLREDUCE=allreduce, HSA_ENABLE_IPC_MODE_LEGACY="0")
    if allreduce == "direct" and backend == "gloo":       # several ranks on one device: the stand-in librccl
        os.environ["CALIB_RCCL_LIBRARY"] = STANDIN
    os.environ.update(extraEnv or {})
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
        sse, P, iters, trace = distributed.refineDistributed(model, P0, offs, s, m, 60, checkEvery=checkEvery)
        np.savez(os.path.join(outDir, f"r{rank}.npz"), sse=sse, P=P, iters=iters, trace=trace,
                 kind=distributed.refineDistributed.lastAllReduce)
    finally:
        dist.destroy_process_group()


def _run(tmp_path, world, backend, case, allreduce="torch", extraEnv=None, checkEvery=8):
    import torch.multiprocessing as mp
    if allreduce == "direct" and backend == "gloo" and not os.path.exists(STANDIN):
        pytest.fail(f"{STANDIN} is missing: __graft_entry__.build() (make -C tests/fake_rccl) builds it")
    mp.spawn(_worker, args=(world, _freePort(), backend, case, allreduce, str(tmp_path), extraEnv, checkEvery),
             nprocs=world, join=True)
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
                                                  (2, "g3", "peer"), (3, "g3", "peer"), (4, "fisheye3", "peer"),
                                                  (2, "g3", "direct"), (3, "g3", "direct"), (4, "g3", "direct"),
                                                  (4, "fisheye3", "direct")])
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


def test_in_library_allreduce_ranks_stop_together(tmp_path):
    """calib_lm_run_sharded(check_every > 0) with several participants: g3 converges after 8 of the 60 allowed
    iterations; looking at the (replicated) done flag every 2 rounds, every 8 rounds, or never must end every rank
    on the same round with the same bits -- what src/calibrate.py:161-168 needs of a sharded run (one decision)."""
    runs = {}
    for ce in (2, 8, 0):
        d = tmp_path / f"ce{ce}"
        d.mkdir()
        runs[ce] = _run(d, 3, "gloo", "g3", "direct", checkEvery=ce)
    for ce, outs in runs.items():
        for o in outs:
            assert str(o["kind"]) == "direct"
            assert int(o["iters"]) == int(runs[0][0]["iters"]) and np.array_equal(o["P"], runs[0][0]["P"]), ce
            assert np.array_equal(o["trace"], runs[0][0]["trace"]), ce
    g = loadGolden("g3_unittest15.npz")
    assert np.abs(runs[2][0]["P"][:10] - g["Pfinal"][:10]).max() < 1e-9


def test_in_library_allreduce_failed_selftest_falls_back_on_every_rank(tmp_path):
    """Rank 1's start-up self-test sees a wrong sum (the stand-in corrupts that rank's first all-reduce): that rank
    aborts its communicator, the MIN all-reduce of the ok flags makes EVERY rank drop the carrier, and the run goes
    through torch.distributed.all_reduce -- same answer, nobody hangs (distributed.directAllReduce)."""
    outs = _run(tmp_path, 3, "gloo", "g3", "direct", extraEnv={"CALIB_STANDIN_CORRUPT": "1:1"})
    (sseR, PR, itR, trR), L = _single("g3")
    for o in outs:
        assert str(o["kind"]) == "torch"
        assert np.array_equal(o["P"], outs[0]["P"]) and int(o["iters"]) == int(outs[0]["iters"])
    assert np.abs(outs[0]["P"][:L] - PR[:L]).max() <= 1e-9 * max(1.0, np.abs(PR[:L]).max())


def _joinWorker(rank, world, port, outDir):
    """calib_rccl_* alone: a rank that never calls ncclCommInitRank must not hang the others."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0",
                      CALIB_STANDIN_JOIN_TIMEOUT="4")
    import time
    import torch.distributed as dist
    import camera_calibration_amd as cca
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = {"error": "", "seconds": 0.0, "allreduce_error": ""}
    try:
        eng = cca.RefineEngine("radtan", "f64")
        eng.rcclLoad(STANDIN)
        ids = [eng.rcclUniqueId() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        if rank != 1:
            t0 = time.perf_counter()
            try:
                eng.rcclInit(world, rank, ids[0], 1.5)
            except RuntimeError as e:
                out["error"] = str(e)
            out["seconds"] = time.perf_counter() - t0
            try:                                          # and the handle has no communicator afterwards
                eng.rcclSelfTest(1.0)
            except RuntimeError as e:
                out["allreduce_error"] = str(e)
        dist.barrier()
        time.sleep(3.5)          # let the abandoned ncclCommInitRank calls run into the stand-in's own timeout
        eng.close()
    finally:
        np.savez(os.path.join(outDir, f"j{rank}.npz"), **out)
        dist.destroy_process_group()


def test_in_library_allreduce_missing_rank_meets_the_deadline(tmp_path):
    import torch.multiprocessing as mp
    mp.spawn(_joinWorker, args=(3, _freePort(), str(tmp_path)), nprocs=3, join=True)
    outs = [np.load(os.path.join(tmp_path, f"j{r}.npz")) for r in range(3)]
    for r in (0, 2):
        assert "did not return before the deadline" in str(outs[r]["error"])
        assert 1.0 < float(outs[r]["seconds"]) < 4.0
        assert "calib_rccl_init has not been called" in str(outs[r]["allreduce_error"])
    assert str(outs[1]["error"]) == ""


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
