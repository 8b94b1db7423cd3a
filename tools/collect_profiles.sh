#!/bin/bash
# After tools/profile_round.sh came back through gpurun: copy the summaries into profiles/ under the round's name and
# rebuild profiles/pmc_fused.json.   bash tools/collect_profiles.sh r03
set -e
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
rnd=${1:-r03}
for w in c3 c2 c4 c5 c3_s20; do
  f=$(ls -t $(find gpurun_out/prof/stats_$w -name "*kernel_stats.csv") | head -1); cp $f profiles/${rnd}_${w}_kernel_stats.csv      # the newest: gpurun merges into what earlier rounds left here
done
for w in c2 c4 c5 c5_full c3_s20 c3 c3_nostream c5_nostream; do cp gpurun_out/prof/bench_$w.json profiles/${rnd}_${w}_bench.json; done
cp gpurun_out/prof/rehearsal_n2.json profiles/${rnd}_rehearsal_n2_gloo_one_gpu.json
cp gpurun_out/prof/rehearsal_n4.json profiles/${rnd}_rehearsal_n4_gloo_one_gpu.json
for n in 2 4; do [ -f gpurun_out/prof/rehearsal_n${n}_direct.json ] && cp gpurun_out/prof/rehearsal_n${n}_direct.json profiles/${rnd}_rehearsal_n${n}_direct.json; done
cp gpurun_out/prof/ubench.txt profiles/${rnd}_ubench.txt
cp gpurun_out/prof/stream_stamps.txt profiles/${rnd}_stream_stamps.txt
# 64-lane batches per fused launch: derived from the launch shape the profiled build reported (config.fused_form in the bench line)
python3 tools/make_pmc_fused.py --round $rnd --workload c3 --tag c3 --model fisheye > /dev/null
python3 tools/make_pmc_fused.py --round $rnd --workload c5 --tag c5 --model radtan > /dev/null
python3 tools/make_pmc_fused.py --round $rnd --workload c2 --tag c2 --model _c2 > /dev/null
python3 tools/make_pmc_fused.py --round $rnd --workload c4 --tag c4 --model _c4 > /dev/null
python3 - <<'PY'
import json
p = "profiles/pmc_fused.json"
d = json.load(open(p))
for k in ("_c2", "_c4"):
    d["by_model"].pop(k, None)
json.dump(d, open(p, "w"), indent=1, sort_keys=True)
PY
echo "profiles/${rnd}_* refreshed"
