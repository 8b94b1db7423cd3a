"""What the peer exchange adds to an LM round, measured without a second GPU: two handles of one process,
each with half of a shard on its own stream, run K rounds (a) independently (no exchange), (b) exchanging
inside their reduce kernels (calib_peer_*, slot memory in the same HBM). The difference is the kernel-side
cost of the exchange (stores, polling, lockstep skew) -- an xGMI hop adds its one-way latency on top.
usage: python tools/peer_cost.py [workload] [views_per_handle] [rounds]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import camera_calibration_amd as cca
from camera_calibration_amd import synthetic


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
    views = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
    K = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    cfg = dict(synthetic.CONFIGS[wl])
    shards = [synthetic.makeShard(cfg, viewStart=r * views, numViews=views, noiseSigma=0.1) for r in range(2)]
    opts = dict(lamInit=1e-3, lamMin=0.0, lamMax=float("inf"), errMin=-float("inf"))
    engs = []
    for sh in shards:
        e = cca.RefineEngine(cfg["model"], cfg["dtype"])
        e.setProblem(sh["viewOffsets"], sh["sensorPoints"], sh["modelPoints"])
        engs.append(e)

    def timed(sharded):
        best = []
        for rep in range(7):
            for e, sh in zip(engs, shards):
                e.lmBegin(sh["P0"], K, **opts)
            for e in engs:                      # bootstrap round
                (e.lmRunSharded if sharded else e.lmRun)(1, 0)
            for e in engs:
                e.lmDone()
            t0 = time.perf_counter()
            for e in engs:
                (e.lmRunSharded if sharded else e.lmRun)(K, 0)
            for e in engs:
                e.lmDone()
            best.append((time.perf_counter() - t0) / K)
            for e in engs:
                e.lmEnd()
        return float(np.median(best))

    solo = timed(False)
    handles = [e.peerPrepare(2, r) for r, e in enumerate(engs)]
    for e in engs:
        e.peerConnect(handles, 20.0)
    peer = timed(True)
    print(f"{wl}: 2 handles x {views} views on one GPU, {K} rounds: independent {solo * 1e3:.4f} ms/round, "
          f"exchanging {peer * 1e3:.4f} ms/round, difference {1e6 * (peer - solo):+.2f} us/round")
    for e in engs:
        e.peerShutdown()
        e.close()


if __name__ == "__main__":
    main()
