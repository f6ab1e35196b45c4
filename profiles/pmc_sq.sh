#!/bin/bash
# SQ stall / instruction-mix counters for the iteration's kernels (separate --pmc passes, kernel-trace only).
cd "$(dirname "$0")/.."; export TMPDIR=/tmp; mkdir -p gpurun_out
i=0; dbs=""
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_VMEM SQ_INSTS_FLAT SQ_INSTS_SMEM" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1)); rm -rf /tmp/pmc_$i
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc_$i -o p -- python bench.py --steps 8 --warmup 3 --cpu-iters 0 > /dev/null 2> gpurun_out/pmc_sq_$i.err
  db=$(find /tmp/pmc_$i -name '*.db' | head -1); dbs="$dbs $db"
done
python profiles/summarize_pmc_db.py "${1:-_kernel}" $dbs > gpurun_out/pmc_sq.txt 2>&1
cat gpurun_out/pmc_sq.txt | head -150
