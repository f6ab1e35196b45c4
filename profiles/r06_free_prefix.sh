#!/bin/bash
# N1 (render_img): a-priori prefix of the frame WITHOUT depth guidance (256 uniform samples): tiles decoded tile-parallel before the on-demand kernel
cd "$(dirname "$0")/.."; REPO=$PWD; export TMPDIR=/tmp; OUT=$REPO/gpurun_out/r06_free_prefix; mkdir -p $OUT
for p in 1 2 3 4 1 2 3; do
  echo -n "office0 MNE_FREE_PREFIX=$p: " | tee -a $OUT/lines.txt
  MNE_FREE_PREFIX=$p timeout 300 python bench.py --mode render_img --steps 40 --warmup 10 --pretrain 100 --cpu-iters 0 --no-variants 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d.get('roofline',{})
print(round(d['ms_per_step'],2), 'ms/pair frac', round(r.get('frac',0),3), 'decoded', d.get('config',{}).get('decoded_samples_per_pair'), 'L1', d.get('config',{}).get('depth_l1_vs_gt'))" | tee -a $OUT/lines.txt
done
