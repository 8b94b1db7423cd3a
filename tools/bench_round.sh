#!/bin/bash
# bench lines of every BASELINE.json shard shape (run on the GPU box through gpurun)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
set -e
for w in c2 c4 c5; do
  python3 bench.py --workload $w --no-cpu-baseline > gpurun_out/bench_$w.json 2> gpurun_out/bench_$w.err
  echo "bench $w done"
done
python3 bench.py --workload c5 --views 1000000 --no-cpu-baseline --steps 20 --warmup 2 > gpurun_out/bench_c5_full.json 2> gpurun_out/bench_c5_full.err
echo "bench c5 full done"
python3 bench.py > gpurun_out/bench_c3.json 2> gpurun_out/bench_c3.err
echo "bench c3 done"
