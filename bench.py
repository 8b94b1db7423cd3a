#!/usr/bin/env python3
"""Benchmark of the LM refinement hot path (BASELINE.json metric: LM iterations/s and
point-residuals/s on a synthetic checkerboard dataset).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = ONE LM iteration (src/calibrate.py:143-168) over the rank's resident views:
per-point residual + Jacobian blocks contracted to per-view J^T J / J^T r with MFMA (fused
kernel; `--lm-mode two_kernel` materialises the compact J in HBM between a jacobian and a gram
kernel instead), per-view Schur elimination, (N > 1: one RCCL all-reduce of the reduce
buffer), accept/reject + L x L solve, back-substitution. Termination tests are disabled for the timed region
(lam_min = 0, lam_max = inf, err_min = -inf) so that exactly K iterations execute, each
with full work. `value` is WEAK scaling (every rank holds `views` views of the config); at N > 1 the line also
carries a `strong` block -- the config's GLOBAL problem (c3: 10 000 views, c5: 1 000 000) split N ways -- and, in
`exchange`, the same K steps timed with every carrier of the per-round sum that passes its self-test (the library's
own ncclAllReduce, torch.distributed.all_reduce; with `--allreduce all` also the in-kernel peer exchange); `value` is
taken with RCCL.
One JSON line is printed by rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_MATRIX_PEAK_TFLOPS = 78.6  # public MI355X sheet; fp64 MFMA = fp64 vector rate
FP32_MATRIX_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 at the fp32 vector rate
# fp64 VALU flops the fused kernel executes per 64-point wave batch (all 64 lanes counted) and its HBM traffic
# are MEASURED quantities of a particular build: rocprofv3 --pmc passes (tools/pmc_pass.sh) summarised into
# profiles/pmc_fused.json by tools/pmc_summary.py together with the commit they were taken at. bench.py quotes
# them with that source; it does not re-measure counters (gpurun forbids mixing --pmc with the timed run).
PMC_FILE = os.path.join(ROOT, "profiles", "pmc_fused.json")


def codeOnly(text):
    """C++ source without // comments, /* */ comments and white space: what a digest of 'the same kernels' should see"""
    import re
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    return re.sub(r"\s+", "", text)


def kernelSourcesDigest():
    import hashlib
    h = hashlib.sha256()
    try:
        for f in ("kernels.hpp", "point_model.hpp"):      # the device code, comments and white space aside
            h.update(codeOnly(open(os.path.join(ROOT, "camera-calibration_amd", "csrc", f)).read()).encode())
    except OSError:
        return None
    return h.hexdigest()[:16]


def countersMatchBuild():
    """do the quoted PMC counters (profiles/pmc_fused.json) come from the kernel sources of THIS tree?"""
    try:
        return json.load(open(PMC_FILE)).get("kernel_sources_sha256_16") == kernelSourcesDigest()
    except Exception:
        return None


def counterEntry(workload):
    """the whole profiles/pmc_fused.json entry of a workload ({} when there is none)"""
    try:
        return json.load(open(PMC_FILE)).get(workload) or {}
    except Exception:
        return {}


def measuredCounters(workload, model):
    """-> (valu fp64 flops per 64-lane batch or None, fused kernel HBM bytes per launch or None,
    jacobian kernel HBM bytes per launch or None, source string)"""
    try:
        d = json.load(open(PMC_FILE))
    except Exception:
        return None, None, None, None
    e = d.get(workload) or {}
    flops = e.get("valu_f64_flops_per_batch") or (d.get("by_model", {}).get(model) or {}).get("valu_f64_flops_per_batch")
    src = e.get("source") or (d.get("by_model", {}).get(model) or {}).get("source")
    return flops, e.get("fused_hbm_bytes_per_launch"), e.get("jacobian_hbm_bytes_per_launch"), src


def algorithmicBytesPerPoint(L, wordBytes):
    """jacobian kernel reads (u,v,X,Y,Z) = 5w, writes the 2 x C block = 2Cw and
    r = 2w -> (7 + 2C) w. gram kernel reads (2C + 2) w."""
    C = L + 6
    return (7 + 2 * C) * wordBytes, (2 * C + 2) * wordBytes


def cpuBaseline(shard, cfgName, seconds=20.0, denseOnly=False):
    """The oracle (numpy restatement of the reference algorithm: dense J, dense J^T J, explicit
    inv) timed on this box's host cores, on a bounded sample of the same workload."""
    from oracle import calib_oracle as orc
    model = orc.RADTAN if shard["model"] == "radtan" else orc.FISHEYE
    N = shard["pointsPerView"]
    views = max(4, min(150, 30000 // N))
    offs = shard["viewOffsets"][:views + 1]
    n = int(offs[-1])
    s, m = shard["sensorPoints"][:n], shard["modelPoints"][:n]
    L = orc.numShared(model)
    P0 = np.concatenate((shard["P0"][:L], shard["P0"][L:L + 6 * views]))
    iters = 2
    t0 = time.perf_counter()
    orc.refineDense(model, P0, offs, s, m, iters, lamMin=0.0, lamMax=np.inf, errMin=-np.inf)
    tDense = time.perf_counter() - t0
    t0 = time.perf_counter()
    orc.refineSchur(model, P0, offs, s, m, iters, lamMin=0.0, lamMax=np.inf, errMin=-np.inf)
    tSchur = time.perf_counter() - t0
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
    except Exception:
        threads = os.cpu_count()
    cPort = None
    try:
        from oracle import c_oracle
        if c_oracle.available() and not denseOnly:
            # the C/OpenMP restatement (block-arrow / Schur form) on ALL host cores, on the whole shard
            full = (shard["viewOffsets"], shard["sensorPoints"], shard["modelPoints"])
            c_oracle.refine(model, shard["P0"], *full, 1, lamMin=0.0, lamMax=np.inf, errMin=-np.inf)    # threads up
            t0 = time.perf_counter()
            c_oracle.refine(model, shard["P0"], *full, 3, lamMin=0.0, lamMax=np.inf, errMin=-np.inf)
            t1 = (time.perf_counter() - t0) / 3
            citers = int(max(2, min(400, 0.75 * seconds / max(t1, 1e-3))))      # ~15 s of host work
            t0 = time.perf_counter()
            c_oracle.refine(model, shard["P0"], *full, citers, lamMin=0.0, lamMax=np.inf, errMin=-np.inf)
            tC = time.perf_counter() - t0
            cPort = {"value": int(full[0][-1]) * citers / tC, "unit": "point-residuals/s",
                     "cores": os.cpu_count(), "kind": "port",
                     "sample": f"whole {cfgName} shard ({int(full[0][-1])} points), {citers} LM iterations of "
                               f"oracle/calib_oracle.c (Schur form, OpenMP over views, {os.cpu_count()} "
                               f"hardware threads), {tC:.1f} s"}
    except Exception as e:       # the C oracle is optional test infrastructure
        cPort = {"error": str(e)}
    return {
        "value": n * iters / tDense, "unit": "point-residuals/s", "cores": int(threads), "kind": "port",
        "sample": f"first {views} views x {N} pts of {cfgName} ({n} points), {iters} LM iterations of "
                  f"the reference's dense algorithm (dense J, J.T@J, inv) restated in numpy "
                  f"(oracle.refineDense), {tDense:.1f} s; BLAS uses {threads} threads, the rest is 1 thread",
        "schur_form_value": n * iters / tSchur,
        "schur_form_note": f"same sample through the block-arrow/Schur numpy oracle, {tSchur:.2f} s",
        "c_openmp_port": cPort,
        # quoted, not measured here: the reference itself in the survey container (BASELINE.md section 2,
        # 8 vCPU Xeon 2.1 GHz, numpy 2.2.6 / sympy 1.14 through the import harness)
        "reference_measured_in_survey": {
            "config1_10x54_radtan": {"s_per_iter": 0.339, "point_residuals_per_s": 1.6e3},
            "config2_scale_1000x54_radtan": {"s_per_iter": 510.0, "point_residuals_per_s": 105.0},
            "configs3to5": "infeasible (dense J = 1.9 TB / 52 TB / 8.4 PB)"},
    }


def apiEndToEnd(cca, synthetic, shard, cfg, workload, local, noise):
    """The drop-in API above the C-ABI, end to end and split by stage (not part of `value`; rank 0, N = 1):
    * `Calibrator.refineCalibrationParameters(A, W, k, allDetections, 50)` -- the reference's own call
      (src/calibrate.py:117-118) -- from a LIST of per-view (sensor, model) arrays of this workload: compose P on the
      device, pack the list (src/calibrate.py:277-282's vstack, once), upload, LM with the reference's stop rule,
      decompose; point-residuals/s = points x iterations executed / wall time of the whole call;
    * `Calibrator.refinePacked` on the c5 per-GPU shard (125 000 views x 88 points, already stacked arrays): compare /
      upload / LM."""
    out = {}
    Model = cca.FisheyeModel if cfg["model"] == "fisheye" else cca.RadialTangentialModel
    cal = cca.Calibrator(Model(), dtype=cfg["dtype"], device=local)
    try:
        offs, s, m = shard["viewOffsets"], shard["sensorPoints"], shard["modelPoints"]
        dets = [(s[a:b], m[a:b]) for a, b in zip(offs[:-1], offs[1:])]
        A0, W0, k0 = cal._decomposeParameterVector(shard["P0"])
        cal.refineCalibrationParameters(A0, W0[:8], k0, dets[:8], 2)          # engine creation, code objects
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            sse, A, W, k = cal.refineCalibrationParameters(A0, W0, k0, dets, 50)
            wall = time.perf_counter() - t0
            if best is None or wall < best[0]:
                best = (wall, dict(cal.lastSeconds), sse)
        wall, sec, sse = best
        iters = int(sec.pop("iters"))
        MN = int(offs[-1])
        out["refineCalibrationParameters"] = {
            "workload": f"{workload}: list of {len(dets)} views x {shard['pointsPerView']} pts, maxIters 50, the reference's stop rule",
            "wall_ms": wall * 1e3, "iterations_executed": iters, "final_sse": sse,
            "value": MN * iters / wall, "unit": "point-residuals/s",
            "stage_ms": {k_: v * 1e3 for k_, v in sec.items()},
            "host_overhead_ms": (wall - sec.get("lm", 0.0) - sec.get("upload", 0.0)) * 1e3,
            "best_of": 3}
    finally:
        cal.close()
    try:
        c5 = dict(synthetic.CONFIGS["c5"])
        sh = synthetic.makeShard(c5, viewStart=0, numViews=c5["views"] // 8, noiseSigma=noise, device=local)
        cal5 = cca.Calibrator(cca.RadialTangentialModel(), dtype="f64", device=local)
        try:
            args5 = (sh["P0"], sh["viewOffsets"], sh["sensorPoints"], sh["modelPoints"], 30)
            cal5.refinePacked(*args5)                                           # first call: engine + pinned staging
            t0 = time.perf_counter()
            _, _, it5, _ = cal5.refinePacked(*args5)
            wall = time.perf_counter() - t0
            sec = dict(cal5.lastSeconds)
            sec.pop("iters", None)
            t0 = time.perf_counter()
            _, _, it5b, _ = cal5.refinePacked(*args5, sameProblem=True)
            wallSame = time.perf_counter() - t0
            MN5 = int(sh["viewOffsets"][-1])
            out["refinePacked_c5_shard"] = {
                "workload": f"c5 per-GPU shard: {c5['views'] // 8} views x {sh['pointsPerView']} pts, stacked arrays, maxIters 30",
                "wall_ms": wall * 1e3, "iterations_executed": int(it5), "value": MN5 * int(it5) / wall,
                "unit": "point-residuals/s", "stage_ms": {k_: v * 1e3 for k_, v in sec.items()},
                "host_overhead_ms": (wall - sec.get("lm", 0.0) - sec.get("upload", 0.0)) * 1e3,
                "upload_GBps": MN5 * 40 / sec["upload"] / 1e9 if sec.get("upload") else None,
                "same_problem_call": {"wall_ms": wallSame * 1e3, "value": MN5 * int(it5b) / wallSame,
                                      "what": "sameProblem=True: the caller vouches for unchanged arrays, no compare, no upload"}}
        finally:
            cal5.close()
    except Exception as e:          # the c5 shard needs ~1.5 GB of host memory and ~10 s of pose sampling
        out["refinePacked_c5_shard"] = {"error": str(e)}
    return out


def strongShardRange(totalViews, world, rank):
    """views [start, end) of `rank` when `totalViews` views of one global problem are split over `world` ranks"""
    from camera_calibration_amd import distributed
    return distributed.strongShardRange(totalViews, world, rank)


class Carrier:
    """one way of carrying the per-round sum over the ranks: its own engine (the peer exchange and the in-library
    communicator are properties of a handle), the ShardedLM on top"""
    def __init__(self, name, eng, lm, describe):
        self.name, self.eng, self.lm, self.describe = name, eng, lm, describe


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100, help="LM iterations per timed segment (<= 100)")
    ap.add_argument("--min-seconds", type=float, default=2.0,
                    help="the K-step segment is repeated until the timed phase lasts at least this long")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="c3", help="c2 | c3 | c4 | c5 (BASELINE.json configs[1..4])")
    ap.add_argument("--views", type=int, default=None, help="views per GPU (default: the config's)")
    ap.add_argument("--noise", type=float, default=0.1, help="sensor noise sigma in px")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-api", action="store_true", help="skip the api_end_to_end block (Calibrator-level calls, N = 1 only)")
    ap.add_argument("--scaling", choices=["both", "weak", "strong"], default="both",
                    help="N > 1: 'weak' = every rank holds the config's per-GPU views (this is `value`); 'strong' = the "
                         "config's global problem split N ways (block `strong`); 'both' (default) measures the two")
    ap.add_argument("--allreduce", choices=["auto", "all", "peer", "direct", "torch"], default="auto",
                    help="the one exchange per LM step at N > 1. 'direct': ncclAllReduce issued by the library on its own "
                         "stream; 'torch': torch.distributed.all_reduce on the bound buffer; 'peer': the reduce kernel itself "
                         "sums over the ranks, point-to-point over xGMI through IPC-mapped slot memory (no collective "
                         "launch). 'auto' (default) times the two RCCL carriers -- every one whose start-up self-test passes on "
                         "EVERY rank (block `exchange`) -- and takes `value` with direct if it came up, else torch. 'all' adds the "
                         "peer exchange to `exchange`: it has been exercised with several processes on ONE GPU only, so the "
                         "default run of an 8-GPU node does not stake its line on it")
    ap.add_argument("--lm-mode", default="fused", choices=["fused", "two_kernel"])
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse the "
                         "multi-rank path with several ranks on one GPU)")
    ap.add_argument("--same-device", action="store_true", help="rehearsal: every rank uses GPU 0")
    ap.add_argument("--rccl-library", default=None,
                    help="rehearsal: resolve the in-library carrier's ncclAllReduce from this library instead of PyTorch's "
                         "librccl.so (tests/fake_rccl/librccl_standin.so lets several ranks share one GPU, which the real "
                         "RCCL refuses); with it `--backend gloo --same-device --allreduce direct` drives rccl_direct")
    ap.add_argument("--force-dist", action="store_true",
                    help="rehearsal: drive the sharded (all-reduce per round) path even with one rank")
    args = ap.parse_args()

    # stdout carries ONE JSON line. Libraries write there too (RCCL prints a version banner at communicator
    # creation): everything up to the line goes to stderr, at file-descriptor level.
    sys.stdout.flush()
    stdoutFd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    # the host driver supports dmabuf IPC only: RCCL and the peer exchange need this in every rank's environment
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.rccl_library:
        os.environ["CALIB_RCCL_LIBRARY"] = os.path.abspath(args.rccl_library)

    import torch
    import camera_calibration_amd as cca
    from camera_calibration_amd import distributed, synthetic

    if args.same_device:
        local = 0
    torch.cuda.set_device(local)
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo")
    devTorch = torch.device("cuda", local)

    cfg = dict(synthetic.CONFIGS[args.workload])
    # the configs c4 and c5 are stated for 8 GPUs: their per-GPU shard is an eighth
    perGpuDefault = cfg["views"] // 8 if args.workload in ("c4", "c5") else cfg["views"]
    viewsPerGpu = args.views or perGpuDefault
    lmOpts = dict(lamInit=1e-3, lamMin=0.0, lamMax=float("inf"), errMin=-float("inf"))
    SEG = 100
    if args.steps > SEG:
        sys.exit(f"bench.py: --steps is at most {SEG} (lambda leaves the fp64 range in longer runs with the stop rule off)")
    # HIP events around the dominant kernel's launches inside the timed region, on the stream they are launched on.
    # An event pair keeps a launch from being dispatched back to back with its neighbours (c3: +12 us per LM round when
    # every launch is bracketed), so every 16th launch is timed -- whatever --steps is.
    PROF_EVERY = 16 if args.lm_mode == "fused" else 1

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()      # the engine's own stream was synchronised by the segment's lmEnd

    def makeEngine(shard):
        eng = cca.RefineEngine(cfg["model"], cfg["dtype"], local)
        t0 = time.perf_counter()
        eng.setProblem(shard["viewOffsets"], shard["sensorPoints"], shard["modelPoints"])
        tUp = time.perf_counter() - t0
        eng.setLmMode(args.lm_mode)
        return eng, tUp

    def makeCarriers(shard, eng0, want):
        """-> list of Carrier, the one `value` is taken with first. Every carrier is self-tested and agreed on by all
        ranks before it is used; the first engine is reused by the first carrier that comes up."""
        out = []
        spare = [eng0]

        def engine():
            return spare.pop() if spare else makeEngine(shard)[0]

        if want in ("auto", "all", "direct") and (args.backend == "nccl" or args.rccl_library):
            e = engine()
            ar = distributed.directAllReduce(e)
            if ar is not None:
                out.append(Carrier("rccl_direct", e, distributed.ShardedLM(e, ar), "ncclAllReduce issued by the library on its own stream"))
            else:
                spare.append(e)
                if want == "direct":
                    sys.exit("bench.py --allreduce direct: the in-library all-reduce could not be set up")
        if want in ("auto", "all", "torch"):
            e = engine()
            out.append(Carrier("torch", e, distributed.ShardedLM(e, distributed.torchAllReduce(e, devTorch)),
                               "torch.distributed.all_reduce"))
        if want in ("all", "peer") and world > 1:
            e = engine()
            ar = distributed.peerExchange(e)
            if ar is not None:
                out.append(Carrier("peer", e, distributed.ShardedLM(e, ar),
                                   "summed inside the reduce kernel, point-to-point over xGMI (IPC-mapped slot memory)"))
            else:
                spare.append(e)
                if want == "peer":
                    sys.exit("bench.py --allreduce peer: the peer exchange could not be set up")
        if not out:                                            # e.g. --allreduce peer with one rank: torch's all-reduce
            e = engine()
            out.append(Carrier("torch", e, distributed.ShardedLM(e, distributed.torchAllReduce(e, devTorch)),
                               "torch.distributed.all_reduce"))
        for e in spare:
            e.close()
        return out

    def measure(shard, eng, lm, minSeconds, profile):
        """R timed segments of exactly --steps LM iterations on `eng` (through `lm` when sharded).
        A timed SEGMENT is exactly K = --steps LM iterations of one refinement: lmBegin (upload of P0, state reset)
        and the bootstrap pass that evaluates P0 are issued and drained BEFORE the clock starts, lmEnd (download of
        P and the trace) runs after it stops -- they are per-refinement costs, reported as segment_overhead_ms.
        With the stop rule disabled lambda overflows after ~310 consecutive rejections at the noise floor, so a
        segment holds at most SEG iterations and every segment restarts from the same perturbed P0 (lambda 1e-3,
        src/calibrate.py:142). The segment is repeated R times (R agreed by all ranks from one calibration
        segment, so that the timed phase lasts >= --min-seconds) and the MEDIAN segment is reported."""
        state = {"P": shard["P0"], "iters": 0, "trace": [], "sse": float("nan")}

        def drain():
            eng.synchronize()             # the engine's own stream (a wait, no copy: lmDone's 1 KiB read-back cost ~10 us)
            torch.cuda.synchronize()

        def segment(k, timed=True):
            tb = time.perf_counter()
            if lm is not None:
                lm.begin(shard["P0"], SEG, **lmOpts)          # lmBegin + bootstrap round
            else:
                eng.lmBegin(shard["P0"], SEG, **lmOpts)
                eng.lmRun(1)
            drain()
            if timed:
                barrier()
            t0 = time.perf_counter()
            if lm is not None:
                lm.run(k)
            else:
                eng.lmRun(k)
            drain()
            if timed:
                barrier()
            t1 = time.perf_counter()
            sse_, P_, it_, tr_ = eng.lmEnd()
            t2 = time.perf_counter()
            state["P"], state["sse"] = P_, sse_
            state["iters"] += it_
            state["trace"].append(tr_)
            return t1 - t0, (t0 - tb) + (t2 - t1)

        # Everything slow is done before the warm-up, so that the device is not left idle between the warm-up
        # and the timed region: after >= 20 ms of idleness the GPU's clocks have dropped and the next ~1 ms of
        # work runs slow. So: event pool and collector first, then an untimed spin-up that brings the clocks up
        # (setup, like generating the data), the W warm-up steps, then the timed segments back to back.
        import gc
        if profile:
            eng.profileEnable(True, every=PROF_EVERY)         # creates the event pool (tens of ms, once)
        gc.collect()
        gc.disable()              # no collector pauses inside the timed region (ranks wait for the slowest each round)
        tSpin = time.perf_counter()
        for _ in range(3):
            segment(SEG, timed=False)                          # clock spin-up, ~25 ms of device work
        tSpin = time.perf_counter() - tSpin
        if args.warmup > 0:
            segment(min(args.warmup, SEG), timed=False)
        tCal, _ = segment(args.steps)                          # one calibration segment fixes the repeat count on every rank
        if dist is not None:
            tc = torch.tensor([tCal], dtype=torch.float64, device="cuda")
            dist.all_reduce(tc, op=dist.ReduceOp.MAX)
            tCal = float(tc.item())
        R = int(min(2001, max(5, np.ceil(minSeconds / max(tCal, 1e-6)))))
        R += 1 - R % 2                                         # odd: the median is one of the segments
        itersBefore = state["iters"]
        state["trace"] = []
        if profile:
            eng.profileEnable(True, every=PROF_EVERY)         # counters back to zero (cheap: the pool exists)
        segTimes, segOver = np.zeros(R), np.zeros(R)
        tPhase = time.perf_counter()
        for j in range(R):
            segTimes[j], segOver[j] = segment(args.steps)
        tPhase = time.perf_counter() - tPhase
        gc.enable()
        if dist is not None:                                   # per segment: the slowest rank
            tt = torch.from_numpy(np.concatenate((segTimes, segOver))).to("cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            both = tt.cpu().numpy()
            segTimes, segOver = both[:R], both[R:]
        res = {"elapsed": float(np.median(segTimes)), "R": R, "segTimes": segTimes, "segOver": segOver, "tPhase": tPhase,
               "tSpin": tSpin, "iters": state["iters"] - itersBefore, "total": args.steps * R, "sse": state["sse"], "P": state["P"],
               "trace": np.vstack(state["trace"]) if state["trace"] else np.zeros((0, 5 + eng.L))}
        if profile:
            res["prof"] = [eng.profileRead(i) for i in range(3)]
            eng.profileEnable(False)
        return res

    def globalPoints(MNlocal):
        if dist is None:
            return MNlocal
        n = torch.tensor([MNlocal], dtype=torch.int64, device="cuda")
        dist.all_reduce(n)
        return int(n.item())

    # ------------------------------------------------------------------ main block (= `value`): weak unless --scaling strong
    viewStart = rank * viewsPerGpu
    strongOnly = dist is not None and args.scaling == "strong"
    if strongOnly:
        totalViews = cfg["views"] if args.views is None else args.views * world
        viewStart, v1 = strongShardRange(totalViews, world, rank)
        viewsPerGpu = v1 - viewStart
    t0 = time.perf_counter()
    shard = synthetic.makeShard(cfg, viewStart=viewStart, numViews=viewsPerGpu, noiseSigma=args.noise, device=local)
    tGen = time.perf_counter() - t0
    MNlocal = int(shard["viewOffsets"][-1])
    eng, tUpload = makeEngine(shard)
    ranksSeen = 1
    carriers = []
    if dist is not None:
        carriers = makeCarriers(shard, eng, args.allreduce)
        # how many ranks the process group really has (not WORLD_SIZE): every rank contributes a one
        seen = torch.ones(1, dtype=torch.int32, device=devTorch if args.backend == "nccl" else "cpu")
        dist.all_reduce(seen)
        ranksSeen = int(seen.item())
        eng = carriers[0].eng
    main = carriers[0] if carriers else None
    exchange = {}
    m = measure(shard, eng, main.lm if main else None, args.min_seconds, profile=True)
    if main:
        exchange[main.name] = m["elapsed"] / args.steps * 1e3
    for c in carriers[1:]:                                     # the same K steps with the other carriers
        mc = measure(shard, c.eng, c.lm, min(args.min_seconds, 0.25), profile=False)
        exchange[c.name] = mc["elapsed"] / args.steps * 1e3
    MNglobal = globalPoints(MNlocal)

    # ------------------------------------------------------------------ strong block: the config's global problem / N
    strong = None
    if dist is not None and args.scaling == "both":
        totalViews = cfg["views"] if args.views is None else args.views * world
        v0, v1 = strongShardRange(totalViews, world, rank)
        if v1 - v0 == viewsPerGpu and world * viewsPerGpu == totalViews:
            # the weak shard IS the strong shard (c4 / c5 at N = 8): measured once
            strong = {"note": "identical to the weak block at this N", "value": MNglobal * args.steps / m["elapsed"],
                      "ms_per_step": m["elapsed"] / args.steps * 1e3, "global_views": totalViews, "views_per_gpu": v1 - v0}
        else:
            sshard = synthetic.makeShard(cfg, viewStart=v0, numViews=v1 - v0, noiseSigma=args.noise, device=local)
            seng, _ = makeEngine(sshard)
            scar = makeCarriers(sshard, seng, {"rccl_direct": "direct", "torch": "torch", "peer": "peer"}[main.name])
            ms = measure(sshard, scar[0].eng, scar[0].lm, min(args.min_seconds, 0.3), profile=False)
            sglobal = globalPoints(int(sshard["viewOffsets"][-1]))
            strong = {"value": sglobal * args.steps / ms["elapsed"], "unit": "point-residuals/s",
                      "lm_iters_per_s": args.steps / ms["elapsed"], "ms_per_step": ms["elapsed"] / args.steps * 1e3,
                      "global_views": totalViews, "views_per_gpu": v1 - v0, "global_points": sglobal,
                      "allreduce": scar[0].name, "segments": int(ms["R"]), "valid": bool(ms["iters"] == ms["total"]),
                      "what": f"the config's global problem ({totalViews} views) split over {world} ranks, views [{v0}, {v1}) "
                              "on rank 0; speed-up = this value / the N = 1 line's value"}
            for c in scar:
                c.eng.close()
        strong["scaling"] = "strong"

    # the same upload once more: what a caller pays per calib_set_problem once the handle's pinned staging exists
    # (the first call above also pays for pinning 8 MiB and the first use of the packing kernel)
    t0 = time.perf_counter()
    eng.setProblem(shard["viewOffsets"], shard["sensorPoints"], shard["modelPoints"])
    tUpload2 = time.perf_counter() - t0
    elapsed, R, segTimes, segOver = m["elapsed"], m["R"], m["segTimes"], m["segOver"]
    sse, P, iters, total, trace = m["sse"], m["P"], m["iters"], m["total"], m["trace"]
    (jacMs, jacN), (gramMs, gramN), (fusedMs, fusedN) = m.get("prof", [(0.0, 0)] * 3)
    twoKernelMsPerStep = None
    jacSteps = gramSteps = args.steps
    if rank == 0 and args.lm_mode == "fused":
        # the same shard through the two-kernel path (compact J through HBM), timed for the
        # HBM roofline of the jacobian kernel and the gram kernel; not part of `value`, and not
        # allowed to take the bench line down with it
        try:
            eng2 = cca.RefineEngine(cfg["model"], cfg["dtype"], local)
            eng2.setProblem(shard["viewOffsets"], shard["sensorPoints"], shard["modelPoints"])
            eng2.setLmMode("two_kernel")
            eng2.lmBegin(shard["P0"], 96, **lmOpts)
            eng2.lmRun(80)                    # ~25 ms: clocks up again after the upload above (see the spin-up note)
            eng2.lmDone()
            eng2.profileEnable(True)
            t2 = time.perf_counter()
            eng2.lmRun(10)
            eng2.lmDone()
            twoKernelMsPerStep = (time.perf_counter() - t2) / 10 * 1e3
            jacMs, jacN = eng2.profileRead(0)
            gramMs, gramN = eng2.profileRead(1)
            jacSteps = gramSteps = 10
            eng2.lmEnd()
            eng2.close()
        except Exception as e:
            print(f"bench: two-kernel pass skipped: {e}", file=sys.stderr)

    if rank == 0:
        L = eng.L
        w = 8 if cfg["dtype"] == "f64" else 4
        jacBytes, gramBytes = algorithmicBytesPerPoint(L, w)
        jacAvgMs = jacMs / max(jacN, 1)
        gramAvgMs = gramMs / max(gramN, 1)
        # a launch covers one chunk of whole views: points per launch = points x steps / launches
        jacPts = MNlocal * jacSteps / max(jacN, 1)
        gramPts = MNlocal * gramSteps / max(gramN, 1)
        jacGBs = jacBytes * jacPts / (jacAvgMs * 1e-3) / 1e9 if jacN else None
        gramGBs = gramBytes * gramPts / (gramAvgMs * 1e-3) / 1e9 if gramN else None
        C = L + 6
        gramFlops = (4 * 16 * 16 + 4 * C) * gramPts      # as executed on full 16x16 MFMA tiles + J^T r
        accepted = int(trace[:, 4].sum())
        valuPerBatch, fusedTraffic, jacTraffic, pmcSource = measuredCounters(args.workload, cfg["model"])
        jacRoof = {"kernel": "jacobian_kernel (two-kernel mode)", "bound": "hbm",
                   "achieved": jacGBs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                   "frac": (jacGBs / HBM_PEAK_GBS) if jacGBs else None, "traffic": jacTraffic,
                   "traffic_source": pmcSource if jacTraffic else None,
                   "algorithmic_bytes_per_launch": jacBytes * jacPts, "points_per_launch": jacPts,
                   "launches_per_step": jacN / jacSteps,
                   "avg_launch_ms": jacAvgMs, "launches_timed": jacN}
        if fusedN:
            fusedAvgMs = fusedMs / fusedN
            sec = fusedAvgMs * 1e-3
            fusedPts = MNlocal                                 # live points: one launch = every point of the shard once
            f64 = cfg["dtype"] == "f64"
            peak = FP64_MATRIX_PEAK_TFLOPS if f64 else FP32_MATRIX_PEAK_TFLOPS
            # `frac` is SURVEY 8(d)'s contract: the matrix flops of J^T J and J^T r, 4 C^2 + 4 C per LIVE point (C = 16 / 15),
            # over the launch's duration, against the dense matrix peak of the dtype. Three more labelled figures beside
            # it, none of them built from flops of dead lanes: what the matrix instructions actually execute (the block
            # form builds the symmetric half: 5 x 512 flop per 4 points; the tile forms 2 x 2048), the contract flops plus
            # the fp64 VALU flops of the per-point model (a PMC measurement, per live point), and the HBM side.
            contractFlops = (4 * C * C + 4 * C) * fusedPts
            stream = eng.fusedForm()[0] > 0
            blockForm = f64 and (stream or shard["pointsPerView"] > 128)
            mfmaExecuted = (5 * 512 // 4 if blockForm else 4 * 16 * 16) * fusedPts
            valuUseful = valuPerBatch * fusedPts / 64.0 if (valuPerBatch and f64) else None
            tfl = lambda fl: fl / sec / 1e12
            mfmaName = "v_mfma_f32_16x16x4_f32" if not f64 else ("v_mfma_f64_4x4x4_4b_f64" if blockForm else "v_mfma_f64_16x16x4_f64")
            kname = "fused_stream_kernel" if stream else "fused_kernel"
            ce = counterEntry(args.workload)
            share, waves = eng.fusedForm()
            # the counters describe a launch of a given shape: same build (digest of the device sources) AND same launch
            # width (share / waves depend on the CU count and on CALIB_STREAM_WAVES)
            sameShape = (ce.get("fused_form") or {}) == {"share": share, "waves": waves} if ce.get("fused_form") else None
            simdCycles = 4 * 256 * sec * 2.4e9                   # 256 CUs x 4 SIMDs at the 2.4 GHz the chip reports under load
            mfmaBusy = ce.get("mfma_busy_cycles_per_launch")
            mainRoof = {"kernel": kname + " (jacobian blocks + " + mfmaName + " J^T J, J on-chip)",
                        "bound": "mfma", "achieved": tfl(contractFlops), "peak": peak, "unit": "TFLOP/s",
                        "frac": tfl(contractFlops) / peak,
                        "frac_is": "SURVEY 8(d): (4 C^2 + 4 C) matrix flops per live point / launch duration / dense matrix peak",
                        "frac_executed": tfl(mfmaExecuted) / peak,
                        "frac_mfma_plus_valu_useful": (tfl(contractFlops + valuUseful) / peak) if valuUseful else None,
                        "hbm_frac": 5 * w * fusedPts / sec / 1e9 / HBM_PEAK_GBS,
                        "traffic": fusedTraffic, "traffic_source": pmcSource if fusedTraffic else None,
                        "counters_are_of_this_build": countersMatchBuild(),
                        "counters_are_of_this_launch_shape": sameShape,
                        "mfma_busy_frac_counters": (mfmaBusy / simdCycles) if mfmaBusy else None,
                        "mfma_busy_frac_counters_is": "SQ_VALU_MFMA_BUSY_CYCLES per launch (PMC pass, profiles/pmc_fused.json) / (1024 SIMDs x this "
                                                      "run's launch duration x 2.4 GHz): the matrix pipe's own busy time, independent of any flop count",
                        "lds_bank_conflict_frac_counters": ce.get("lds_bank_conflict_frac"),
                        "wait_inst_any_frac_counters": ce.get("wait_inst_any_frac"),
                        "mfma_flops_per_launch": contractFlops, "mfma_flops_executed_per_launch": mfmaExecuted,
                        "valu_fp64_flops_per_launch_live_points": valuUseful,
                        "valu_flops_source": pmcSource if valuPerBatch else "none: profiles/pmc_fused.json has no entry",
                        "sustained_whole_chip_TFLOPs_measured": {"v_mfma_f64_16x16x4": 46.4, "v_mfma_f64_4x4x4_4b": 75.8,
                                                                 "v_fma_f64": 61.8, "source": "profiles/r02_ubench.txt"},
                        "points_per_launch": fusedPts,
                        "algorithmic_hbm_bytes_per_launch": 5 * w * fusedPts,
                        "avg_launch_ms": fusedAvgMs, "launches_timed": fusedN,
                        "timed_every_nth_launch": PROF_EVERY}
        else:
            mainRoof = jacRoof
        out = {
            "metric": "LM iters/sec & residuals/sec, 10k-view checkerboard @1/2/4/8 GPU",
            "value_is": "point-residuals/s = points x LM iterations / wall time of the timed region "
                        "(lm_iters_per_s beside it)",
            "value": MNglobal * args.steps / elapsed,
            "unit": "point-residuals/s",
            "lm_iters_per_s": args.steps / elapsed,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "timing": {"what": "median over `segments` timed segments of exactly `steps` LM iterations each (max over "
                               "ranks per segment); lmBegin, the bootstrap pass and lmEnd are outside the segments "
                               "(round 1 timed ONE region, a mean that included them: `ms_per_step_whole_refinement` is "
                               "that definition)",
                       "segments": int(R), "segment_ms_min": float(segTimes.min() * 1e3),
                       "segment_ms_median": float(np.median(segTimes) * 1e3), "segment_ms_max": float(segTimes.max() * 1e3),
                       "segment_overhead_ms": float(np.median(segOver) * 1e3),
                       "ms_per_step_whole_refinement": float(np.mean(segTimes + segOver) / args.steps * 1e3),
                       "ms_per_refinement": float(np.mean(segTimes + segOver) * 1e3),
                       "timed_phase_s": float(m["tPhase"])},
            "higher_is_better": True, "scaling": "strong" if strongOnly else "weak", "vs_baseline": None,
            "dtype": cfg["dtype"], "data": "synthetic",
            "config": {"workload": f"{args.workload}: {viewsPerGpu} views x {shard['pointsPerView']} pts per GPU, "
                                   f"{cfg['model']}, {cfg['dtype']}, sensor noise {args.noise} px",
                       "views_per_gpu": viewsPerGpu, "points_per_view": shard["pointsPerView"],
                       "global_points": MNglobal, "distortion": cfg["model"], "parallelism": f"views-sharded x{world}",
                       "allreduce": main.describe if main else None,
                       "allreduce_used_for_value": main.name if main else None,
                       "ranks_seen": ranksSeen,
                       "exchange_selftest": None if main is None else ("torch.distributed.all_reduce needs none" if main.name == "torch"
                                                                        else "exact sums of rank-dependent values over all ranks, against a deadline, passed on every rank"),
                       "lm_mode": args.lm_mode,
                       "fused_form": dict(zip(("share", "waves"), eng.fusedForm()))},
            "exchange": None if dist is None else {"ms_per_step": exchange, "what": "the same K steps of the weak block with every "
                                                   "carrier of the per-round sum that passed its self-test on every rank "
                                                   "(the peer exchange only with --allreduce all | peer)",
                                                   "backend": args.backend},
            "strong": strong,
            "roofline": mainRoof,
            "roofline_jacobian_kernel": jacRoof,
            "two_kernel_ms_per_step": twoKernelMsPerStep,
            "roofline_gram": {"kernel": "gram_kernel (two-kernel mode, v_mfma_f64_16x16x4_f64)", "bound": "hbm",
                              "achieved": gramGBs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": (gramGBs / HBM_PEAK_GBS) if gramGBs else None,
                              "mfma_tflops": gramFlops / (gramAvgMs * 1e-3) / 1e12 if gramN else None,
                              "mfma_peak_tflops": FP64_MATRIX_PEAK_TFLOPS,
                              "mfma_util": gramFlops / (gramAvgMs * 1e-3) / 1e12 / FP64_MATRIX_PEAK_TFLOPS if gramN else None,
                              "superseded_by": "`roofline` (the fused kernel): J never reaches HBM there and the same J^T J runs at "
                                               "`roofline.frac` of the matrix peak; this kernel re-reads a materialised J and is bound by "
                                               "the memory system (4 flop/B), which is what its low MFMA utilisation says",
                              "points_per_launch": gramPts, "launches_per_step": gramN / gramSteps,
                              "avg_launch_ms": gramAvgMs, "launches_timed": gramN},
            "valid": bool(iters == total),
            "lm": {"accepted_steps_in_timed_region": accepted, "final_sse": sse,
                   "iterations_executed": int(iters), "iterations_requested": int(total),
                   "note": "sensor noise makes the estimate differ from the generating parameters (statistical "
                           "error); parity with the reference is asserted on noise-free data in tests/",
                   "max_rel_err_intrinsics_vs_truth": float(np.max(np.abs(P[:L] - shard["Ptrue"][:L])
                                                                   / np.maximum(np.abs(shard["Ptrue"][:L]), 1.0)))},
            "setup_s": {"generate": tGen, "pack_upload": tUpload, "pack_upload_second_call": tUpload2,
                        "pack_upload_second_call_GBps": (MNlocal * 40 / tUpload2 / 1e9) if tUpload2 > 0 else None,
                        "clock_spinup_300_untimed_iterations": m["tSpin"]},
        }
        if iters != total:
            # with the stop rule disabled lambda still overflows to inf after ~310 consecutive rejections
            # (x10 each): the loop then ends early and later rounds are no-ops -- such a run is not a measurement
            print(f"bench: only {iters} of {total} LM iterations executed (lambda left the fp64 range); "
                  f"use fewer --steps", file=sys.stderr)
        if world == 1 and not args.no_api and args.lm_mode == "fused":
            try:
                out["api_end_to_end"] = apiEndToEnd(cca, synthetic, shard, cfg, args.workload, local, args.noise)
            except Exception as e:
                out["api_end_to_end"] = {"error": str(e)}
        if not args.no_cpu_baseline:
            # N = 1: the dense numpy sample and the C/OpenMP port on all host cores (~17 s). N > 1: rank 0 times the
            # dense sample only (a few seconds, the other ranks wait in the closing barrier) -- the host's cores are
            # shared by N ranks there, and the N = 1 line of the same node holds the full baseline.
            try:
                out["cpu_baseline"] = cpuBaseline(shard, args.workload, denseOnly=world > 1)
                if world > 1:
                    out["cpu_baseline"]["note"] = ("rank 0 only, dense sample only; the C/OpenMP port on all host cores is in "
                                                   "the N = 1 line")
            except Exception as e:
                out["cpu_baseline"] = {"error": str(e)}
        sys.stdout.flush()
        os.dup2(stdoutFd, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    for c in carriers:
        c.eng.close()
    if not carriers:
        eng.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
