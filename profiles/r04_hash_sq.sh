#!/bin/bash
# SQ counters of the hash-grid table update's kernels (stand-alone calls after 40 pipeline steps; one pass, kernel-trace only)
cd "$(dirname "$0")/.."; REPO=$PWD; export TMPDIR=/tmp; OUT=$REPO/gpurun_out/r04_hash_sq; mkdir -p $OUT
cd /tmp
rm -rf /tmp/pm1; timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES SQ_BUSY_CYCLES -d /tmp/pm1 -o p -- python $REPO/profiles/r04_hash_ablate.py main > $OUT/p1.log 2>&1
python $REPO/profiles/pmc_dump.py $(find /tmp/pm1 -name '*.db' | head -1) hash_ | tee $OUT/sq1.txt
rm -rf /tmp/pm2; timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM -d /tmp/pm2 -o p -- python $REPO/profiles/r04_hash_ablate.py main > $OUT/p2.log 2>&1
python $REPO/profiles/pmc_dump.py $(find /tmp/pm2 -name '*.db' | head -1) hash_ | tee $OUT/sq2.txt
tail -3 $OUT/p2.log
