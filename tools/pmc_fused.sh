#!/bin/bash
# SQ counters of the fused kernel on c3, one pass per counter group (no tracing domains besides kernel-trace)
export TMPDIR=/tmp
cd /tmp
rocprofv3 --list-avail > /root/repo/gpurun_out/avail.txt 2>&1
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_IDX_ACTIVE" \
           "SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $grp -d /root/repo/gpurun_out/pmcf_$i -- python3 /root/repo/tools/kexp.py --workload c3 --steps 6 > /root/repo/gpurun_out/pmcf_$i.log 2>&1 || echo "group $i failed"
done
