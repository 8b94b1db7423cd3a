#!/bin/bash
# per-kernel durations of the LM round on three shard shapes (run on the GPU box through gpurun)
export TMPDIR=/tmp
cd /tmp
for w in "c5 --views 125000" "c4 --views 12500" "c3"; do
  tag=$(echo $w | cut -d' ' -f1)
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/kp_$tag -- python3 /root/repo/tools/kexp.py --workload $w > /root/repo/gpurun_out/kp_$tag.log 2>&1 || exit 1
  echo "== $w"
  python3 /root/repo/tools/dbstats.py $(ls -t /root/repo/gpurun_out/kp_$tag/*/*.db | head -1) | grep "schur\|backsub\|fused\|reduce_k"
done
