#!/bin/bash
# SQ counter passes over the default bench (run on the GPU box through gpurun): bash tools/pmc_pass.sh <tag> [bench args...]
# One rocprofv3 --pmc pass per counter group (<= 8 SQ counters each), kernel trace only; summaries by tools/pmc_summary.py.
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
export TMPDIR=/tmp
cd /tmp
G1="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_WAVES SQ_INSTS_VALU_FMA_F64"
G2="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS"
G3="SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH"
G4="SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT64 SQ_THREAD_CYCLES_VALU SQ_BUSY_CU_CYCLES"
n=1
for G in "$G1" "$G2" "$G3" "$G4"; do
  rocprofv3 --kernel-trace --pmc $G --output-format csv -d $R/gpurun_out/pmc_${TAG}_$n -- python3 $R/bench.py --no-cpu-baseline --steps 10 --warmup 2 "$@" > $R/gpurun_out/pmc_${TAG}_$n.log 2>&1
  echo "pass $n done"
  n=$((n+1))
done
