#!/bin/bash
# SQ counters of the fused kernel on c3, one pass per counter group (no tracing domains besides kernel-trace)
export TMPDIR=/tmp
cd /tmp
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM" \
           "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $grp -d /root/repo/gpurun_out/pmcf_$i -- python3 /root/repo/tools/kexp.py --workload ${1:-c3} ${2:+--views $2} --steps 6 > /root/repo/gpurun_out/pmcf_$i.log 2>&1 || echo "group $i failed"
  python3 /root/repo/tools/dbpmc.py $(ls -t /root/repo/gpurun_out/pmcf_$i/*/*.db | head -1) fused
done
