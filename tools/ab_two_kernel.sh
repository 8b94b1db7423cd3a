#!/bin/bash
# Same-box A/B of library builds in TWO-KERNEL mode (jacobian kernel -> compact J in HBM -> gram kernel): VARIANTS="a b" bash tools/ab_two_kernel.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do for v in product $VARIANTS; do
 if [ $v = product ]; then unset CALIB_LM_LIBRARY; else export CALIB_LM_LIBRARY=$R/tools/diag/lib/$v/libcalib_lm.so; fi
 for w in c3 c5; do python3 $R/bench.py --no-cpu-baseline --no-api --workload $w 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); j=d['roofline_jacobian_kernel']; g=d['roofline_gram']; print('$v','$w','two-kernel ms/step',round(d['two_kernel_ms_per_step'],4),'jac us',round(j['avg_launch_ms']*1e3,1),'frac',round(j['frac'],3),'gram us',round(g['avg_launch_ms']*1e3,1),'hbm',round(g['frac'],3),'mfma',round(g['mfma_util'],3))"; done; done; done
