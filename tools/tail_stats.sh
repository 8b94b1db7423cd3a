#!/bin/bash
# Per-kernel average durations of an LM loop (rocprofv3 --kernel-trace --stats over tools/kexp.py), for quick
# A/B experiments on the GPU box:  bash tools/tail_stats.sh TAG [workloads...]
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=${1:-x}; shift
wls=${@:-c3 c4 c2 c5}
export TMPDIR=/tmp KEXP_NOPROF=1
mkdir -p $R/gpurun_out/ts
cd /tmp
for w in $wls; do
  rm -rf $R/gpurun_out/ts/${tag}_$w
  views=""; [ $w = c4 ] && views="--views 12500"; [ $w = c5 ] && views="--views 125000"      # per-GPU shards
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ts/${tag}_$w -- python3 $R/tools/kexp.py --workload $w $views --steps 200 > $R/gpurun_out/ts/${tag}_$w.log 2>&1
  find $R/gpurun_out/ts/${tag}_$w -name "*kernel_trace.csv" -delete
  f=$(find $R/gpurun_out/ts/${tag}_$w -name "*kernel_stats.csv" | head -1)
  echo "== $tag $w: $(grep -o 'ms/iter [0-9.]*' $R/gpurun_out/ts/${tag}_$w.log)"
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n = r["Name"]
    for k in ("fused_kernel", "schur_kernel", "reduce_kernel", "update_backsub"):
        if k in n:
            print(f"   {k:24s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:8.2f} us")
PY
done
