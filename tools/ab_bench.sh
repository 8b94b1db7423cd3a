#!/bin/bash
# Same-box A/B of two builds of the library: bench lines alternate between the product build and
# camera-calibration_amd/lib/<variant>/libcalib_lm.so (CALIB_LM_LIBRARY).  bash tools/ab_bench.sh <variant> [workloads...]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
var=$1; shift
wls=${@:-c3 c5 c4 c2}
for rep in 1 2; do
  for v in product $var; do
    if [ $v = product ]; then unset CALIB_LM_LIBRARY; else export CALIB_LM_LIBRARY=$R/camera-calibration_amd/lib/$var/libcalib_lm.so; fi
    for w in $wls; do
      python3 $R/bench.py --no-cpu-baseline --no-api --workload $w 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', '$w', round(d['ms_per_step'],4), round(d['roofline']['avg_launch_ms']*1e3,2))"
    done
  done
done
