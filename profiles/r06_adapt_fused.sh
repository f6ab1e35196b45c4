#!/bin/bash
# adaptive prefix schedule on the fused launch, driver's form (fresh map): default thresholds vs mode 0 pinned (MNE_NO_ADAPT=1)
cd "$(dirname "$0")/.."; export TMPDIR=/tmp; OUT=gpurun_out/r06_adapt_fused; mkdir -p $OUT
for k in 1 2 3; do for v in 0 1; do
  echo -n "driver form MNE_NO_ADAPT=$v: " | tee -a $OUT/lines.txt
  MNE_NO_ADAPT=$v timeout 300 python bench.py --steps 20 --warmup 5 --cpu-iters 0 --no-variants 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value'],1), 'it/s', {k[:16]: round(v*1000,1) for k,v in r['other_kernels_avg_ms'].items()})" | tee -a $OUT/lines.txt
done; done
for v in 0 1; do
  echo -n "first_frame-like (steps 60 warmup 0) MNE_NO_ADAPT=$v: " | tee -a $OUT/lines.txt
  MNE_NO_ADAPT=$v timeout 300 python bench.py --steps 60 --warmup 0 --cpu-iters 0 --no-variants 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'it/s')" | tee -a $OUT/lines.txt
done
