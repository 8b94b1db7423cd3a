#!/bin/bash
# N ranks on ONE GPU through the DEFAULT multi-GPU carrier (rccl_direct: the library's own ncclAllReduce), resolved from
# tests/fake_rccl's stand-in because the real RCCL refuses two ranks on one device. Writes gpurun_out/prof/rehearsal_n{2,4}_direct.json
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
mkdir -p gpurun_out/prof
for n in 2 4; do
  timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29800 + n)) \
    bench.py --gpus $n --same-device --backend gloo --allreduce auto --rccl-library tests/fake_rccl/librccl_standin.so \
    --steps 20 --warmup 5 --min-seconds 0.3 > gpurun_out/prof/rehearsal_n${n}_direct.log 2>&1 || { echo "n=$n FAILED"; tail -20 gpurun_out/prof/rehearsal_n${n}_direct.log; exit 1; }
  grep "^{" gpurun_out/prof/rehearsal_n${n}_direct.log > gpurun_out/prof/rehearsal_n${n}_direct.json
  python3 - gpurun_out/prof/rehearsal_n${n}_direct.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read())
print("n", d["n_gpus"], "carrier", d["config"]["allreduce_used_for_value"], "ranks_seen", d["config"]["ranks_seen"], "ms_per_step", d["ms_per_step"],
      "exchange", d["exchange"]["ms_per_step"], "strong", (d.get("strong") or {}).get("ms_per_step"), "cpu_baseline", (d.get("cpu_baseline") or {}).get("value"))
PY
done
