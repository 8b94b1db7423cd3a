#!/bin/bash
# Round profile refresh, run on the GPU box through gpurun:  bash tools/profile_round.sh
# kernel-trace/stats and the PMC counters in separate passes; outputs under gpurun_out/prof, summarised into profiles/
# afterwards (tools/make_pmc_fused.py, cp of the *_kernel_stats.csv).
# (gpurun allows 20 minutes per call: `bash tools/profile_round.sh pmc` runs the counter passes alone, `... rest` everything else)
R=${GRAFT_REPO_ROOT:-/root/repo}
part=${1:-all}
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/prof
cd /tmp
if [ $part = pmc ] || [ $part = all ]; then
for w in c3 c5 c2 c4; do
  bash $R/tools/pmc_pass.sh $w --workload $w
  python3 $R/tools/pmc_compact.py $w        # raw counter files are tens of MiB per pass: keep the means only
  rm -f $R/gpurun_out/pmc_${w}_*.log
done
fi
[ $part = pmc ] && exit 0
# `bench`: only the bench lines again (after tools/collect_profiles.sh has rebuilt profiles/pmc_fused.json from this build's
# counter passes, so that the lines quote counters of the build and launch shape they ran)
if [ $part != bench ]; then
for w in c3 c2 c4 c5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof/stats_$w -- python3 $R/bench.py --no-cpu-baseline --no-api --workload $w > $R/gpurun_out/prof/stats_$w.log 2>&1
  find $R/gpurun_out/prof/stats_$w -name "*kernel_trace.csv" -delete      # the stats are what is kept (gpurun_out/ returns <= 64 MiB)
  echo "stats $w done"
done
# the driver's own command under the profiler: its fused-kernel average is what BENCH_rNN's roofline must agree with
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof/stats_c3_s20 -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-api > $R/gpurun_out/prof/stats_c3_s20.log 2>&1
find $R/gpurun_out/prof/stats_c3_s20 -name "*kernel_trace.csv" -delete
echo "stats c3 (driver flags) done"
fi
cd $R
for w in c2 c4 c5; do
  python3 bench.py --workload $w --no-cpu-baseline --no-api > gpurun_out/prof/bench_$w.json 2> gpurun_out/prof/bench_$w.err
  echo "bench $w done"
done
python3 bench.py --workload c5 --views 1000000 --no-cpu-baseline --no-api --steps 20 --warmup 2 > gpurun_out/prof/bench_c5_full.json 2> gpurun_out/prof/bench_c5_full.err
echo "bench c5 full done"
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/prof/bench_c3_s20.json 2> gpurun_out/prof/bench_c3_s20.err
echo "bench c3 driver flags done"
python3 bench.py > gpurun_out/prof/bench_c3.json 2> gpurun_out/prof/bench_c3.err
echo "bench c3 done"
CALIB_FUSED_STREAM=0 python3 bench.py --no-cpu-baseline --no-api > gpurun_out/prof/bench_c3_nostream.json 2> gpurun_out/prof/bench_c3_nostream.err
CALIB_FUSED_STREAM=0 python3 bench.py --no-cpu-baseline --no-api --workload c5 > gpurun_out/prof/bench_c5_nostream.json 2> gpurun_out/prof/bench_c5_nostream.err
echo "bench without the stream form done"
[ $part = bench ] && exit 0
# N ranks on ONE GPU over gloo (rehearsal of the N > 1 line: weak + strong blocks, every carrier)
for n in 2 4; do
  timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29700 + n)) \
    bench.py --gpus $n --same-device --backend gloo --allreduce all --no-cpu-baseline --no-api --steps 20 --warmup 5 --min-seconds 0.2 > gpurun_out/prof/rehearsal_n$n.log 2>&1
  grep "^{" gpurun_out/prof/rehearsal_n$n.log > gpurun_out/prof/rehearsal_n$n.json
  echo "rehearsal n=$n done"
done
bash tools/rehearse_direct.sh          # 2 and 4 ranks through the library's own ncclAllReduce (stand-in librccl)
# whole-chip fp64 pipe rates (tools/ubench; binaries are built by hand, see the sources' headers)
for u in ubench6 ubench7 ubench8 ubench10 mfma4x4_layout; do
  [ -x tools/ubench/$u ] && { echo "== $u"; tools/ubench/$u; } >> gpurun_out/prof/ubench.txt 2>&1
done
echo "ubench done"
# per-phase stamps of the stream kernel (diagnostic build: bash tools/diag/build_stream_stamps.sh first)
if [ -f tools/diag/lib/stream_stamps/libcalib_lm.so ]; then
  { python3 tools/diag/stream_stamps.py c3; python3 tools/diag/stream_stamps.py c5 125000; } 2>&1 | grep -vE "amdgpu.ids" > gpurun_out/prof/stream_stamps.txt
  echo "stream stamps done"
fi
