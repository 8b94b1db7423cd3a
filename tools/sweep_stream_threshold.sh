#!/bin/bash
# Where does the stream form of the fused kernel start to pay? bench lines with CALIB_FUSED_STREAM=0 / 1 on c3- and c5-shaped shards of
# 1 000 .. 7 000 views (the default switches at two views per wave slot = 8 192 views).
R=${GRAFT_REPO_ROOT:-/root/repo}
for wl in c3 c5; do for n in 1000 2500 5000 7000; do for v in 0 1; do
  export CALIB_FUSED_STREAM=$v
  python3 $R/bench.py --no-cpu-baseline --no-api --workload $wl --views $n 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$wl views=$n stream=$v ms/step', round(d['ms_per_step'],4), 'fused us', round(d['roofline']['avg_launch_ms']*1e3,2))"
done; done; done
