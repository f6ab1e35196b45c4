#!/bin/bash
# SQ counters of the frame-render kernels, per dispatch (separate --pmc passes, kernel-trace only)
cd "$(dirname "$0")/.."; REPO=$PWD; export TMPDIR=/tmp; OUT=$REPO/gpurun_out/r05_render_sq; mkdir -p $OUT
cd /tmp
i=0; dbs=""
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_VMEM SQ_INSTS_FLAT SQ_INSTS_SMEM"; do
  i=$((i+1)); rm -rf /tmp/pmc_rs_$i
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc_rs_$i -o p -- python $REPO/bench.py --mode render_img --steps 40 --warmup 20 --pretrain 20 > /dev/null 2> $OUT/pmc_$i.err
  dbs="$dbs $(find /tmp/pmc_rs_$i -name '*.db' | head -1)"
done
python $REPO/profiles/r05_sq_dispatches.py "RenderArgs" 4 $dbs 2>&1 | grep "Li0EE\|, 0>\|decode_kernel" | cut -c1-220 > $OUT/sq_dispatches.txt
cat $OUT/sq_dispatches.txt
