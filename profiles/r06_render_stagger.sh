#!/bin/bash
# N1 (render_img): the waves of a SIMD started a fraction of a tile apart (MNE_FRAME_STAGGER = start delay per wave slot, x ~1024 cycles)
cd "$(dirname "$0")/.."; REPO=$PWD; export TMPDIR=/tmp; OUT=$REPO/gpurun_out/r06_render_stagger; mkdir -p $OUT
line() {  # label, env
  echo -n "$1: " | tee -a $OUT/lines.txt
  env $2 timeout 300 python bench.py --mode render_img --steps 40 --warmup 10 --pretrain 100 --cpu-iters 0 --no-variants 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d.get('roofline',{})
print(round(d['ms_per_step'],2), 'ms/pair frac', round(r.get('frac',0),3), 'decoded', d.get('config',{}).get('decoded_samples_per_pair'), 'L1', d.get('config',{}).get('depth_l1_vs_gt'))" | tee -a $OUT/lines.txt
}
for s in 0 2 5 10 15 20 30 0 10; do line "office0 stagger $s" "MNE_FRAME_STAGGER=$s"; done
