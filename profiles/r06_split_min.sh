#!/bin/bash
# tile_adam_kernel: split threshold of long lists (MNE_TILE_SPLIT_MIN; split = max(it, 2 * entries / 2048)) on the list-bound workloads
cd "$(dirname "$0")/.."; REPO=$PWD; export TMPDIR=/tmp; OUT=$REPO/gpurun_out/r06_split_min; mkdir -p $OUT
for cfg in indoor scannet; do
for sm in 4096 1024 2048 3072 6144 8192 4096; do
  echo -n "$cfg MNE_TILE_SPLIT_MIN=$sm: " | tee -a $OUT/lines.txt
  MNE_TILE_SPLIT_MIN=$sm timeout 300 python bench.py --config $cfg --no-variants --cpu-iters 0 --steps 300 --warmup 50 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value'],1), 'it/s', round(d['ms_per_step'],4), 'ms; tile_adam', round(r['avg_launch_ms']*1000,1), 'us')" | tee -a $OUT/lines.txt
done; done
python profiles/r05_indoor_lists.py 2>/dev/null | tail -12 | tee $OUT/indoor_lists.txt
