#!/bin/bash
# fused_stream_kernel: launch width sweep (CALIB_STREAM_WAVES) on one workload.  bash tools/sweep_stream_waves.sh c3 2048 4096 ...
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
w=$1; shift
export CALIB_FUSED_STREAM=1
for n in "$@"; do
  export CALIB_STREAM_WAVES=$n
  python3 $R/bench.py --no-cpu-baseline --no-api --workload $w $BENCH_EXTRA 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('waves=$n', '$w', 'ms/step', round(d['ms_per_step'],4), 'fused us', round(d['roofline']['avg_launch_ms']*1e3,2))"
done
