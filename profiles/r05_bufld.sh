#!/bin/bash
# plane corner rows through buffer loads (one VALU of address arithmetic per load instead of three): frame renders and the training iteration
cd "$(dirname "$0")/.."; REPO=$PWD; export TMPDIR=/tmp; OUT=$REPO/gpurun_out/r05_bufld; mkdir -p $OUT
for v in main bufld main bufld; do
  echo -n "render_img $v: " | tee -a $OUT/lines.txt
  timeout 300 python profiles/r03_variant_bench.py $v --mode render_img --steps 80 --warmup 20 --pretrain 100 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), 'ms/pair')" | tee -a $OUT/lines.txt
done
for cfg in office0 scannet indoor; do
for v in main bufld main bufld; do
  echo -n "$cfg training $v: " | tee -a $OUT/lines.txt
  timeout 300 python profiles/r03_variant_bench.py $v --config $cfg --no-variants --cpu-iters 0 --steps 300 --warmup 50 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'it/s', round(d['ms_per_step'],4), 'ms')" | tee -a $OUT/lines.txt
done; done
