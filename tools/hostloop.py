"""Host cost per LM round of the sharded loop (world size 1, tiny problem so the device is never the limit)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import torch.distributed as dist
import camera_calibration_amd as cca
from camera_calibration_amd import distributed, synthetic
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29519")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
sh = synthetic.makeShard("c2", numViews=64, noiseSigma=0.1)
eng = cca.RefineEngine("radtan", "f64")
eng.setProblem(sh["viewOffsets"], sh["sensorPoints"], sh["modelPoints"])
ar = distributed.torchAllReduce(eng, torch.device("cuda", 0))
opts = dict(lamInit=1e-3, lamMin=0.0, lamMax=float("inf"), errMin=-float("inf"))
N, SEGS = 100, 20          # segments: with the stop rule off lambda overflows after ~300 rejections
def run(name, body):
    host = total = 0.0
    for _ in range(SEGS):
        eng.lmBegin(sh["P0"], N + 8, **opts)
        eng.lmLocal(); ar(); eng.lmUpdate()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(N):
            body()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        eng.lmEnd()
        host += t1 - t0; total += t2 - t0
    print(f"{name:34s} host {1e6*host/(N*SEGS):7.1f} us/round   incl. drain {1e6*total/(N*SEGS):7.1f} us/round")
def full(): eng.lmLocal(); ar(); eng.lmUpdate()
def noar(): eng.lmLocal(); eng.lmUpdate()
buf = ar.buffer
def aronly(): dist.all_reduce(buf)
run("lmLocal + all_reduce + lmUpdate", full)
run("lmLocal + lmUpdate", noar)
t0 = time.perf_counter()
for _ in range(2000): aronly()
torch.cuda.synchronize()
print(f"{'all_reduce only':34s} {1e6*(time.perf_counter()-t0)/2000:7.1f} us/call")
eng.setStream(None)
host = total = 0.0
for _ in range(SEGS):
    eng.lmBegin(sh["P0"], N + 8, **opts)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); eng.lmRun(N); t1 = time.perf_counter(); eng.lmDone(); t2 = time.perf_counter(); eng.lmEnd()
    host += t1 - t0; total += t2 - t0
print(f"{'C loop calib_lm_run':34s} host {1e6*host/(N*SEGS):7.1f} us/round   incl. drain {1e6*total/(N*SEGS):7.1f} us/round")
dist.destroy_process_group()
