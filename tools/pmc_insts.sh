#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_WAVES SQ_INSTS_BRANCH --output-format csv -d $R/gpurun_out/pmc_${TAG}_1 -- python3 $R/bench.py --no-cpu-baseline --steps 10 --warmup 2 --min-seconds 0.05 "$@" > $R/gpurun_out/pmc_${TAG}_1.log 2>&1
find $R/gpurun_out/pmc_${TAG}_1 -name "*kernel_trace.csv" -delete
