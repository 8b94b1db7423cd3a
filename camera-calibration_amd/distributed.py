"""Views sharded over the GPUs of one node: one process per GPU, one all-reduce per LM step.

Rows of J belonging to different views are independent given P, and a view's six
extrinsics appear only in its own rows (src/jacobian.py:81-83), so each rank keeps a
contiguous range of views resident in its HBM and eliminates their 6x6 blocks locally.
What is shared is the L x L Schur system: per LM round every rank contributes its partial
sums (reduce buffer of calib_lm.h: 444 doubles for L = 10, 364 for L = 9) to ONE sum over the ranks, after
which every rank takes the identical accept/reject decision and solves the identical L x L system
redundantly. No other data-path exchange. The sum has three carriers: the library's own ncclAllReduce
(directAllReduce, the default), torch.distributed.all_reduce on the bound buffer (torchAllReduce, the
fallback), and -- opt-in, CALIB_ALLREDUCE=peer -- the engine's reduce kernel exchanging its elements
point-to-point through IPC-mapped slot memory (peerExchange; exercised between processes on one GPU only).

The driver is generic over the shard engine (RefineEngine on the GPU; tests drive the
same code with a CPU test double over gloo).
"""
import numpy as np


def partitionViews(viewOffsets, worldSize):
    """Contiguous view ranges balanced by point count -> list of (viewStart, viewEnd) per rank.
    Every rank gets at least one view when there are at least worldSize views; with fewer views the
    surplus ranks get empty ranges (an empty shard contributes zeros and follows every decision)."""
    offs = np.asarray(viewOffsets, dtype=np.int64)
    M = offs.shape[0] - 1
    total = int(offs[-1])
    bounds = [0]
    for r in range(1, worldSize):
        target = total * r / worldSize
        v = int(np.searchsorted(offs, target, side="left"))
        # pick the boundary whose point offset is closest to the target
        if v > 0 and abs(int(offs[v - 1]) - target) <= abs(int(offs[min(v, M)]) - target):
            v -= 1
        if M >= worldSize:
            lo, hi = bounds[-1] + 1, M - (worldSize - r)      # leave a view for this rank and for each later one
        else:
            lo, hi = bounds[-1], M
        bounds.append(min(max(v, lo), hi))
    bounds.append(M)
    return [(bounds[r], bounds[r + 1]) for r in range(worldSize)]


def strongShardRange(totalViews, worldSize, rank):
    """Strong scaling of ONE uniform global problem: views [start, end) of `rank` when `totalViews` views are dealt
    out to `worldSize` ranks in contiguous ranges whose sizes differ by at most one."""
    totalViews, worldSize, rank = int(totalViews), int(worldSize), int(rank)
    if worldSize < 1 or not 0 <= rank < worldSize or totalViews < 0:
        raise ValueError("strongShardRange: need 0 <= rank < worldSize and totalViews >= 0")
    return totalViews * rank // worldSize, totalViews * (rank + 1) // worldSize


def validateGlobalProblem(viewOffsets):
    """What a single engine would reject, checked on the GLOBAL offsets every rank holds, so that all ranks
    raise together instead of one of them leaving the others inside the per-round all-reduce."""
    offs = np.asarray(viewOffsets, dtype=np.int64)
    if offs.ndim != 1 or offs.shape[0] < 2:
        raise ValueError("no views to refine")
    if offs[0] != 0 or np.any(np.diff(offs) < 0):
        raise ValueError("view_offsets must start at 0 and be non-decreasing")
    if np.any(np.diff(offs) == 0):
        raise np.linalg.LinAlgError("a view without points makes J^T J + lambda diag(J^T J) singular")


def shardProblem(P, viewOffsets, sensorPoints, modelPoints, L, viewRange):
    """Cut one rank's views out of a global problem -> (P_local, offsets_local, sensor, model)."""
    v0, v1 = viewRange
    offs = np.asarray(viewOffsets, dtype=np.int64)
    a, b = int(offs[v0]), int(offs[v1])
    P = np.asarray(P, dtype=np.float64).ravel()
    Pl = np.concatenate((P[:L], P[L + 6 * v0:L + 6 * v1]))
    sensor = None if sensorPoints is None else np.asarray(sensorPoints)[a:b]
    return Pl, offs[v0:v1 + 1] - a, sensor, np.asarray(modelPoints)[a:b]


class ShardedLM:
    """One rank of a sharded LM run: engine.lmLocal -> allReduce -> engine.lmUpdate per round."""

    def __init__(self, eng, allReduce):
        self.eng = eng
        self.allReduce = allReduce      # callable(): sums the bound reduce buffer over ranks, in place
        self.maxIters = 0

    def begin(self, Plocal, maxIters, **lmOptions):
        self.maxIters = int(maxIters)
        self.eng.lmBegin(Plocal, maxIters, **lmOptions)
        self.round()                    # round 0: evaluate P0, first step

    def round(self):
        if getattr(self.allReduce, "inLibrary", False):
            self.eng.lmRunSharded(1, 0)
            return
        self.eng.lmLocal()
        self.allReduce()
        self.eng.lmUpdate()

    def run(self, rounds, checkEvery=0):
        """`rounds` LM iterations; with checkEvery > 0 the (replicated, hence identical on every
        rank) done flag is read every checkEvery rounds and the loop stops early."""
        if getattr(self.allReduce, "inLibrary", False):
            # the library all-reduces between its local and update steps itself: whole rounds from C
            self.eng.lmRunSharded(int(rounds), int(checkEvery))
            return
        for i in range(int(rounds)):
            self.round()
            if checkEvery > 0 and (i + 1) % checkEvery == 0 and i + 1 < rounds and self.eng.lmDone():
                break

    def end(self):
        return self.eng.lmEnd()


def torchAllReduce(eng, device):
    """Bind a float64 CUDA tensor as the engine's reduce buffer and order the engine's kernels on
    torch's current stream, so torch.distributed.all_reduce (RCCL) needs no host sync."""
    import torch
    import torch.distributed as dist
    buf = torch.zeros(eng.reduceSize(), dtype=torch.float64, device=device)
    eng.setStream(torch.cuda.current_stream(device).cuda_stream)
    eng.bindReduceBuffer(buf.data_ptr())

    def allReduce():
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)

    allReduce.buffer = buf      # keep the storage alive
    return allReduce


def _allRanksOk(ok, dev):
    """MIN over the ranks of a local success flag: every rank takes the same branch."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(t.item())


def peerExchange(eng, timeoutSeconds=60.0, selfTestRounds=64, selfTestTimeout=10.0):
    """Sum the reduce buffer over the ranks INSIDE the engine's reduce kernel (include/calib_lm.h,
    calib_peer_*): every rank maps every other rank's slot memory through HIP IPC, stores its elements
    point-to-point over xGMI and polls its own memory for the others' -- no collective launch per LM round.
    The 64-byte IPC handles travel over the default process group (any backend); before the exchange is
    trusted it is run on known values against a deadline.
    -> the all-reduce callable (inLibrary: whole rounds run from C), or None when any rank could not set it
    up (the caller then falls back to directAllReduce / torchAllReduce); every rank takes the same branch."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = torch.device("cuda", getattr(eng, "device", torch.cuda.current_device()))
    mine = None
    try:
        mine = eng.peerPrepare(world, rank)
    except Exception:       # noqa: BLE001 -- agreed on below
        mine = None
    handles = [None] * world
    dist.all_gather_object(handles, mine)
    ok = all(hd is not None for hd in handles)
    if ok:
        try:
            eng.peerConnect(handles, timeoutSeconds)
        except Exception:   # noqa: BLE001
            ok = False
    if _allRanksOk(ok, dev):
        try:
            eng.peerSelfTest(selfTestRounds, selfTestTimeout)
        except Exception:   # noqa: BLE001
            ok = False
        ok = _allRanksOk(ok, dev)
    else:
        ok = False
    if not ok:
        try:
            eng.peerShutdown()
        except Exception:   # noqa: BLE001
            pass
        return None

    def allReduce():
        raise RuntimeError("the peer exchange happens inside lmRunSharded")

    allReduce.inLibrary = True
    allReduce.kind = "peer"
    return allReduce


def rcclLibraryPath():
    """librccl.so the in-library all-reduce resolves ncclAllReduce & co. from: the one PyTorch ships (and has already
    loaded), unless CALIB_RCCL_LIBRARY names another build -- the GPU tests put tests/fake_rccl's stand-in there to
    run several ranks on ONE device, which the real RCCL refuses."""
    import os
    import torch
    return os.environ.get("CALIB_RCCL_LIBRARY") or os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")


def directAllReduce(eng, timeoutSeconds=30.0, libraryPath=None, initTimeoutSeconds=120.0):
    """Let the engine issue the all-reduce itself (ncclAllReduce on its own stream, RCCL resolved from
    the librccl.so PyTorch ships and has already loaded): no hand-off to the process group's stream,
    and whole LM rounds run from C. The communicator is bootstrapped over the default process group (any backend:
    only the 128-byte unique id travels over it); before it is trusted it all-reduces a rank-dependent vector and
    checks the sums against a deadline.
    -> the all-reduce callable, or None when any rank could not set it up (the caller then uses
    torchAllReduce); every rank takes the same branch."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = torch.device("cuda", getattr(eng, "device", torch.cuda.current_device()))      # the ENGINE's device

    def allRanksOk(ok):
        return _allRanksOk(ok, dev)

    ok = True
    try:
        eng.rcclLoad(libraryPath or rcclLibraryPath())
    except Exception:
        ok = False
    if not allRanksOk(ok):
        return None
    ids = [None]
    if rank == 0:
        try:
            ids[0] = eng.rcclUniqueId()
        except Exception:
            ids[0] = None
    dist.broadcast_object_list(ids, src=0)
    if ids[0] is None:
        return None
    ok = True
    try:
        eng.rcclInit(world, rank, ids[0], initTimeoutSeconds)          # collective
        eng.rcclSelfTest(timeoutSeconds)
    except Exception:
        ok = False
    if not allRanksOk(ok):
        try:
            eng.rcclShutdown()
        except Exception:
            pass
        return None

    def allReduce():
        eng.lmAllReduce()

    allReduce.inLibrary = True
    allReduce.kind = "direct"
    return allReduce


def _raiseTogether(dist, torch, err, device):
    """MIN-reduce an ok flag: a rank whose local step failed re-raises its error, every other rank raises too --
    nobody is left waiting in a later collective."""
    ok = torch.tensor([0 if err is not None else 1], dtype=torch.int32, device=device)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if err is not None:
        raise err
    if int(ok.item()) == 0:
        raise RuntimeError("refineDistributed: another rank rejected its shard; see that rank's error")


def refineDistributed(modelName, P0, viewOffsets, sensorPoints, modelPoints, maxIters, dtype="f64",
                      checkEvery=8, engineFactory=None, allReduceFactory=None, **lmOptions):
    """Refine ONE global problem with the views sharded over the ranks of the default
    torch.distributed process group (every rank passes the same global arrays and gets the same
    global result back). -> (sse, P (K,), iters, trace)

    engineFactory(viewOffsets, sensor, model) / allReduceFactory(engine) exist so that the protocol
    can be exercised without GPUs (tests); by default the shard engine is a RefineEngine on
    cuda:LOCAL_RANK and the reduction a torch.distributed.all_reduce (RCCL) on its stream."""
    import os
    import torch
    import torch.distributed as dist
    from . import engine as engine_mod
    rank, world = dist.get_rank(), dist.get_world_size()
    L = engine_mod.NUM_SHARED[engine_mod.MODEL_IDS[modelName]]
    validateGlobalProblem(viewOffsets)                 # identical on every rank: everybody raises, or nobody
    if int(maxIters) <= 0:
        raise UnboundLocalError("local variable 'Pt_error' referenced before assignment (maxIters=0)")
    parts = partitionViews(viewOffsets, world)
    Pl, ol, sl, ml = shardProblem(P0, viewOffsets, sensorPoints, modelPoints, L, parts[rank])
    onGpu = engineFactory is None
    okDevice = "cpu"
    eng = allReduce = None
    err = None
    try:
        if onGpu:
            local = int(os.environ.get("LOCAL_RANK", rank))
            torch.cuda.set_device(local)
            if dist.get_backend() == "nccl":
                okDevice = torch.device("cuda", local)
            eng = engine_mod.RefineEngine(modelName, dtype, local)
            eng.setProblem(ol, sl, ml)
        else:
            eng = engineFactory(ol, sl, ml)
    except Exception as e:           # noqa: BLE001 -- re-raised below, after every rank knows
        err = e
    _raiseTogether(dist, torch, err, okDevice)
    if onGpu:
        # CALIB_ALLREDUCE = auto (default) | direct | torch | peer. auto = RCCL: the library's own ncclAllReduce
        # (whole rounds from C), else torch.distributed.all_reduce. peer = the exchange inside the reduce kernel,
        # point-to-point through IPC-mapped slot memory: OPT-IN, because it has only ever run between processes
        # that share one GPU (no xGMI link crossed). Each carrier is self-tested and agreed
        # on by all ranks before it is used; a carrier that does not come up falls through to the next.
        want = os.environ.get("CALIB_ALLREDUCE", "auto")
        if want == "peer" and world > 1:
            allReduce = peerExchange(eng)
        # (auto takes the in-library carrier only under the nccl backend: a gloo group means ranks that may share a
        # device, where RCCL cannot come up; an explicit "direct" is tried under any backend)
        if allReduce is None and (want == "direct" or (want in ("auto", "peer") and dist.get_backend() == "nccl")):
            allReduce = directAllReduce(eng)
        if allReduce is None:
            allReduce = torchAllReduce(eng, torch.device("cuda", eng.device))
    else:
        allReduce = allReduceFactory(eng)
    lm = ShardedLM(eng, allReduce)
    # begin = lmBegin (may reject this rank's shard) + round 0 (first collective): the rejection is agreed on
    # before anybody enters the collective
    try:
        lm.maxIters = int(maxIters)
        eng.lmBegin(Pl, maxIters, **lmOptions)
    except Exception as e:           # noqa: BLE001
        err = e
    _raiseTogether(dist, torch, err, okDevice)
    lm.round()
    lm.run(maxIters, checkEvery=checkEvery)
    sse, Plocal, iters, trace = lm.end()
    refineDistributed.lastAllReduce = getattr(allReduce, "kind", "torch")
    # assemble the global parameter vector: shared part is replicated, extrinsics are gathered
    gathered = [None] * world
    dist.all_gather_object(gathered, (parts[rank], Plocal[L:]))
    P = np.empty(L + 6 * (len(viewOffsets) - 1))
    P[:L] = Plocal[:L]
    for (v0, v1), ext in gathered:
        P[L + 6 * v0:L + 6 * v1] = ext
    if hasattr(eng, "close"):
        eng.close()
    return sse, P, iters, trace
