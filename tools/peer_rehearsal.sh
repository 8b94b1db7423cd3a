#!/bin/bash
# N ranks on ONE GPU (rehearsal): the per-round exchange carried by torch.distributed (gloo), by the
# library's peer exchange, and -- at one rank -- nothing. Prints the bench lines.
# usage: tools/peer_rehearsal.sh OUTDIR [WORKLOAD] [VIEWS_PER_RANK]
out=${1:-gpurun_out/r2}; wl=${2:-c3}; views=${3:-2500}
mkdir -p "$out"
for n in 2 4; do
  for ar in torch peer; do
    timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 \
      --master-port $((29600 + n)) bench.py --gpus $n --same-device --backend gloo --allreduce $ar \
      --workload $wl --views $views --no-cpu-baseline --no-api --steps 50 > "$out/rehearsal_${wl}_n${n}_${ar}.log" 2>&1 \
      || { echo "n=$n $ar FAILED"; tail -5 "$out/rehearsal_${wl}_n${n}_${ar}.log"; exit 1; }
    python - "$out/rehearsal_${wl}_n${n}_${ar}.log" $n $ar <<'PY'
import json, sys
for ln in open(sys.argv[1]):
    if ln.startswith("{"):
        d = json.loads(ln)
        print(f"n={sys.argv[2]} {sys.argv[3]:5s} ms_per_step={d['ms_per_step']:.4f} value={d['value']:.3e} allreduce={d['config'].get('allreduce')}")
PY
  done
done
