#!/bin/bash
# Counter passes over a short bench run (on the GPU box through gpurun): bash tools/pmc_pass.sh <tag> [bench args...]
# One rocprofv3 --pmc pass per counter group (<= 8 SQ counters, FETCH_SIZE and WRITE_SIZE on their own), kernel trace
# only; summaries by tools/pmc_summary.py / tools/make_pmc_fused.py.
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
export TMPDIR=/tmp
cd /tmp
G1="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_WAVES SQ_INSTS_VALU_FMA_F64"
G2="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS"
G3="SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH"
G4="SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_TRANS_F32"
G5="FETCH_SIZE"
G6="WRITE_SIZE"
n=1
for G in "$G1" "$G2" "$G3" "$G4" "$G5" "$G6"; do
  rocprofv3 --kernel-trace --pmc $G --output-format csv -d $R/gpurun_out/pmc_${TAG}_$n -- python3 $R/bench.py --no-cpu-baseline --no-api --steps 10 --warmup 2 --min-seconds 0.05 "$@" > $R/gpurun_out/pmc_${TAG}_$n.log 2>&1
  find $R/gpurun_out/pmc_${TAG}_$n -name "*kernel_trace.csv" -delete
  echo "pmc $TAG pass $n done"
  n=$((n+1))
done
