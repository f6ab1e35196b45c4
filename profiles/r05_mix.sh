#!/bin/bash
# tile_adam dispatch order: sorted (mix0) vs light/heavy alternation behind the first 256 (main) / 512 items
cd "$(dirname "$0")/.."; REPO=$PWD; export TMPDIR=/tmp; OUT=$REPO/gpurun_out/r05_mix; mkdir -p $OUT
for cfg in office0 scannet indoor; do
  for v in main mix0 mix512 main mix0; do
    echo -n "$cfg $v: " | tee -a $OUT/lines.txt
    timeout 300 python profiles/r03_variant_bench.py $v --config $cfg --no-variants --cpu-iters 0 --steps 300 --warmup 50 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d.get('roofline',{})
print(round(d['value'],1), 'it/s', round(d['ms_per_step'],4), 'ms', 'tile_adam us', round(r.get('launch_us',0) or 0,1), 'frac', round(r.get('frac',0),3))" | tee -a $OUT/lines.txt
  done
done
