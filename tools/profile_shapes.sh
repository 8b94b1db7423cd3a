#!/bin/bash
# rocprofv3 kernel stats of bench.py on the other shard shapes (run on the GPU box through gpurun)
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd /tmp
set -e
for w in c5 c4 c2; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$w -- python3 $R/bench.py --workload $w --no-cpu-baseline > $R/gpurun_out/prof_$w.log 2>&1
  echo "stats $w done"
done
