mkdir -p gpurun_out/r2
rm -f gpurun_out/r2/diag.log
timeout -k 10 1100 python -m pytest tests -m gpu -q > gpurun_out/r2/t4.log 2>&1; tail -5 gpurun_out/r2/t4.log
for w in c3 c5 c4 c2; do
  V=""; [ $w = c5 ] && V="--views 125000"; [ $w = c4 ] && V="--views 12500"
  echo "== $w" >> gpurun_out/r2/diag.log
  KEXP_NOPROF=1 timeout -k 10 120 python tools/kexp.py --workload $w $V --steps 200 2>&1 | grep -v amdgpu.ids >> gpurun_out/r2/diag.log
done
cat gpurun_out/r2/diag.log
cd /tmp && export TMPDIR=/tmp
for w in c4; do
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2/prof_$w -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --workload $w > $GRAFT_REPO_ROOT/gpurun_out/r2/prof_$w.log 2>&1
done
echo prof done
