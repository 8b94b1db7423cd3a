#!/bin/bash
# Round profile refresh, run on the GPU box through gpurun:  bash tools/profile_round.sh
# kernel-trace/stats and the PMC counters in separate passes; outputs under gpurun_out/, summarised into profiles/
# afterwards (tools/make_pmc_fused.py, cp of the *_kernel_stats.csv).
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/prof
cd /tmp
for w in c3 c2 c4 c5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof/stats_$w -- python3 $R/bench.py --no-cpu-baseline --workload $w > $R/gpurun_out/prof/stats_$w.log 2>&1
  find $R/gpurun_out/prof/stats_$w -name "*kernel_trace.csv" -delete      # the stats are what is kept (gpurun_out/ returns <= 64 MiB)
  echo "stats $w done"
done
bash $R/tools/pmc_pass.sh c3
bash $R/tools/pmc_pass.sh c5 --workload c5
cd $R
for w in c2 c4 c5; do
  python3 bench.py --workload $w --no-cpu-baseline > gpurun_out/prof/bench_$w.json 2> gpurun_out/prof/bench_$w.err
  echo "bench $w done"
done
python3 bench.py --workload c5 --views 1000000 --no-cpu-baseline --steps 20 --warmup 2 > gpurun_out/prof/bench_c5_full.json 2> gpurun_out/prof/bench_c5_full.err
echo "bench c5 full done"
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/prof/bench_c3_s20.json 2> gpurun_out/prof/bench_c3_s20.err
echo "bench c3 driver flags done"
python3 bench.py > gpurun_out/prof/bench_c3.json 2> gpurun_out/prof/bench_c3.err
echo "bench c3 done"
# whole-chip fp64 pipe rates (tools/ubench/ubench6-8; binaries are built by hand, see the sources' headers)
for u in ubench6 ubench7 ubench8 mfma4x4_layout; do
  [ -x tools/ubench/$u ] && { echo "== $u"; tools/ubench/$u; } >> gpurun_out/prof/ubench.txt 2>&1
done
echo "ubench done"
# per-phase stamps and in-situ ablation of the fused kernel (diagnostic builds: bash tools/diag/build_diag.sh first)
if [ -f tools/diag/lib/stamps/libcalib_lm.so ]; then
  { for a in "c3" "c5 125000" "c2" "c4 12500"; do python3 tools/diag/fused_stamps.py $a; done
    for a in "c3" "c5 --views 125000" "c2" "c4 --views 12500"; do bash tools/diag/fused_ablation.sh $a; done; } 2>&1 | grep -vE "amdgpu.ids" > gpurun_out/prof/fused_diag.txt
  echo "fused diag done"
fi
