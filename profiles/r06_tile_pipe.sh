#!/bin/bash
# tile_adam_kernel: entries and rows of pass k + 1 requested under pass k (TILE_PIPE=1, the tree) vs every pass on its own (variant pipe0)
cd "$(dirname "$0")/.."; REPO=$PWD; export TMPDIR=/tmp; OUT=$REPO/gpurun_out/r06_tile_pipe; mkdir -p $OUT
for cfg in indoor scannet office0 apartment; do
for v in pipe0 main pipe0 main; do
  echo -n "$cfg $v: " | tee -a $OUT/lines.txt
  timeout 300 python profiles/r03_variant_bench.py $v --config $cfg --no-variants --cpu-iters 0 --steps 300 --warmup 50 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value'],1), 'it/s', round(d['ms_per_step'],4), 'ms; tile_adam', round(r['avg_launch_ms']*1000,1), 'us')" | tee -a $OUT/lines.txt
done; done
