#!/bin/bash
# Same-box A/B of an environment knob of the library: bench lines alternate between unset and NAME=VALUE.
#   bash tools/ab_env.sh NAME VALUE [workloads...]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
name=$1; val=$2; shift 2
wls=${@:-c3 c5 c2 c4}
for rep in 1 2; do
  for v in off on; do
    if [ $v = off ]; then unset $name; else export $name=$val; fi
    for w in $wls; do
      python3 $R/bench.py --no-cpu-baseline --no-api --workload $w 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', '$v', '$w', 'ms/step', round(d['ms_per_step'],4), 'fused us', round(d['roofline']['avg_launch_ms']*1e3,2))"
    done
  done
done
