#!/bin/bash
# Round profile refresh, run on the GPU box through gpurun:  bash tools/profile_round.sh
# (kernel-trace/stats and the PMC counters in separate passes; outputs under gpurun_out/)
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd /tmp
set -e
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c3 -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/prof_c3.log 2>&1
echo "stats pass done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -- python3 $R/bench.py --no-cpu-baseline --steps 10 --warmup 2 > $R/gpurun_out/pmc_fetch.log 2>&1
echo "fetch pass done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -- python3 $R/bench.py --no-cpu-baseline --steps 10 --warmup 2 > $R/gpurun_out/pmc_write.log 2>&1
echo "write pass done"
cd $R
for w in c2 c4 c5; do
  python3 bench.py --workload $w --no-cpu-baseline > gpurun_out/bench_$w.json 2> gpurun_out/bench_$w.err
  echo "bench $w done"
done
python3 bench.py --workload c5 --views 1000000 --no-cpu-baseline --steps 20 --warmup 2 > gpurun_out/bench_c5_full.json 2> gpurun_out/bench_c5_full.err
echo "bench c5 full done"
python3 bench.py > gpurun_out/bench_c3.json 2> gpurun_out/bench_c3.err
echo "bench c3 done"
