#!/bin/bash
# Round 3 counter passes + kernel stats of the default bench workload (variants and CPU baseline off: they launch the same
# kernel names on other shapes).  Separate --pmc passes, --kernel-trace only (gpurun refuses anything else).
cd "$(dirname "$0")/.."; REPO=$PWD; export TMPDIR=/tmp; mkdir -p gpurun_out
B="python $REPO/bench.py --no-variants --cpu-iters 0"
cd /tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_MFMA"; do
  i=$((i+1)); rm -rf /tmp/pmc_$i
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc_$i -o p -- $B --steps 10 --warmup 3 > /dev/null 2> $REPO/gpurun_out/r03_pmc_$i.err
done
db() { find /tmp/pmc_$1 -name '*.db' | head -1; }
python $REPO/profiles/pmc_summary.py $(db 1) $(db 2) $(db 3) $REPO/gpurun_out/r03_pmc_traffic.json $REPO/gpurun_out/r03_pmc_traffic.txt > /dev/null
# kernel stats of the driver-form command and of the 200-step default
for form in "--steps 20 --warmup 5" "--steps 200 --warmup 20"; do
  tag=$(echo $form | tr -d ' -'); rm -rf /tmp/ks_$tag
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ks_$tag -o k -- $B $form > $REPO/gpurun_out/r03_ks_$tag.log 2>&1
  d=$(find /tmp/ks_$tag -name '*.db' | head -1)
  python $REPO/profiles/summarize_rocprof_db.py $d 60 > $REPO/gpurun_out/r03_kernel_stats_$tag.txt 2>&1
  python $REPO/profiles/timeline.py $d 12 40 > $REPO/gpurun_out/r03_timeline_$tag.txt 2>&1
  tail -1 $REPO/gpurun_out/r03_ks_$tag.log | cut -c1-300
done
