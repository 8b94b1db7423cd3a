mkdir -p gpurun_out/r2
rm -f gpurun_out/r2/diag.log
timeout -k 10 1100 python -m pytest tests -m gpu -q -x > gpurun_out/r2/t4.log 2>&1; tail -5 gpurun_out/r2/t4.log
for w in c3 c5 c4 c2; do
  V=""; [ $w = c5 ] && V="--views 125000"; [ $w = c4 ] && V="--views 12500"
  echo "== $w" >> gpurun_out/r2/diag.log
  KEXP_NOPROF=1 timeout -k 10 120 python tools/kexp.py --workload $w $V --steps 200 2>&1 | grep -v amdgpu.ids >> gpurun_out/r2/diag.log
done
cat gpurun_out/r2/diag.log
