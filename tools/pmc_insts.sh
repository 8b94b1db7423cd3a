#!/bin/bash
# Instruction-count counters of a short bench run (one pass): bash tools/pmc_insts.sh <tag> [bench args...]
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_WAVES SQ_INSTS_VALU_INT32 --output-format csv -d $R/gpurun_out/pmc_${TAG}_1 -- python3 $R/bench.py --no-cpu-baseline --no-api --steps 10 --warmup 2 --min-seconds 0.05 "$@" > $R/gpurun_out/pmc_${TAG}_1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_BRANCH --output-format csv -d $R/gpurun_out/pmc_${TAG}_2 -- python3 $R/bench.py --no-cpu-baseline --no-api --steps 10 --warmup 2 --min-seconds 0.05 "$@" > $R/gpurun_out/pmc_${TAG}_2.log 2>&1
find $R/gpurun_out/pmc_${TAG}_1 $R/gpurun_out/pmc_${TAG}_2 -name "*kernel_trace.csv" -delete
