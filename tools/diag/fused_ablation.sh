#!/bin/bash
# In-situ ablation of the fused kernel (diagnostic builds of tools/diag/build_diag.sh): HIP-event time of the
# kernel with parts compiled out.  bash tools/diag/fused_ablation.sh [workload] [extra kexp flags]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
w=${1:-c3}; shift
for v in "" noV noVM noVMW noVMWG; do
  if [ -n "$v" ]; then export CALIB_LM_LIBRARY=$R/tools/diag/lib/$v/libcalib_lm.so; else unset CALIB_LM_LIBRARY; fi
  t=$(python3 $R/tools/kexp.py --workload $w --steps 100 "$@" | grep -o "fused [0-9.]* us")
  case "$v" in
    "") what="the whole kernel";;
    noV) what="without the Jacobian arithmetic";;
    noVM) what="... and without the MFMAs";;
    noVMW) what="... and without the slab stores";;
    noVMWG) what="... and without the global point loads";;
  esac
  echo "$w  $t   $what"
done
