#!/bin/bash
# Same-box A/B of library builds: the product and tools/diag/lib/<variant>/libcalib_lm.so (tools/diag/build_variant.sh),
# bench lines alternating.  VARIANTS="a b" bash tools/ab_variant.sh [workloads...]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
wls=${@:-c3 c5}
for rep in 1 2; do
  for v in product $VARIANTS; do
    if [ $v = product ]; then unset CALIB_LM_LIBRARY; else export CALIB_LM_LIBRARY=$R/tools/diag/lib/$v/libcalib_lm.so; fi
    for w in $wls; do
      python3 $R/bench.py --no-cpu-baseline --no-api --workload $w 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', '$w', 'ms/step', round(d['ms_per_step'],4), 'fused us', round(d['roofline']['avg_launch_ms']*1e3,2))"
    done
  done
done
