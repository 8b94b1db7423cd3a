#!/bin/bash
# Same-box A/B of the stream form of the fused kernel (CALIB_FUSED_STREAM=0 / 1) on the bench workloads, plus
# rocprofv3 kernel stats of both.  bash tools/ab_stream.sh [workloads...]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp
wls=${@:-c3 c5}
mkdir -p $R/gpurun_out/ab
for rep in 1 2; do
  for v in 0 1; do
    export CALIB_FUSED_STREAM=$v
    for w in $wls; do
      python3 $R/bench.py --no-cpu-baseline --no-api --workload $w 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('stream=$v', '$w', 'ms/step', round(d['ms_per_step'],4), 'fused us', round(d['roofline']['avg_launch_ms']*1e3,2))"
    done
  done
done
cd /tmp
for v in 0 1; do
  export CALIB_FUSED_STREAM=$v
  for w in $wls; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ab/stats_${w}_s$v -- python3 $R/bench.py --no-cpu-baseline --no-api --workload $w > $R/gpurun_out/ab/stats_${w}_s$v.log 2>&1
    find $R/gpurun_out/ab/stats_${w}_s$v -name "*kernel_trace.csv" -delete
    echo "== stream=$v $w"; f=$(find $R/gpurun_out/ab/stats_${w}_s$v -name "*kernel_stats.csv" | head -1); head -8 $f | cut -c1-200
  done
done
