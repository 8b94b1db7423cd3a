#!/bin/bash
# update kernel time against the shard's view count, 16 lanes per view (update_backsub_kernel) vs one lane per view
# (update_backsub_lane_kernel): where the switch kUpdLaneViews belongs.   bash tools/sweep_upd_lane.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp KEXP_NOPROF=1
mkdir -p $R/gpurun_out/ts
cd /tmp
for spec in "c3 10000" "c4 12500" "c5 16384" "c5 32768" "c5 65536" "c5 125000"; do
  set -- $spec; w=$1; n=$2
  for lane in 1000000000 1; do
    d=$R/gpurun_out/ts/upd_${w}_${n}_$lane; rm -rf $d
    CALIB_UPD_LANE_VIEWS=$lane rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $R/tools/kexp.py --workload $w --views $n --steps 200 > $d.log 2>&1
    f=$(find $d -name "*kernel_stats.csv" | head -1)
    python3 - "$f" $w $n $lane <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "update_backsub" in r["Name"]:
        print(f"{sys.argv[2]} views {sys.argv[3]:>7s} {'lane' if sys.argv[4] == '1' else '16-lane':8s} {r['Name'].split('(')[0].split('::')[-1][:44]:44s} {float(r['AverageNs'])/1e3:8.2f} us")
PY
    rm -rf $d
  done
done
