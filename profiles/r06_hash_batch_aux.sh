#!/bin/bash
# hash-grid iteration: loss scalars + next batch on a third stream beside the slice launch (default) vs behind the table update (MNE_HASH_BATCH_INLINE=1)
cd "$(dirname "$0")/.."; REPO=$PWD; export TMPDIR=/tmp; OUT=$REPO/gpurun_out/r06_hash_aux; mkdir -p $OUT
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value'],1), 'it/s', round(d['ms_per_step'],4), 'ms;', r['kernel'][:22], round(r['avg_launch_ms']*1000,1), 'us', {k[:14]: round(v*1000,1) for k,v in r['other_kernels_avg_ms'].items()})"; }
for v in 1 0 1 0; do
  echo -n "office0_hash inline=$v: " | tee -a $OUT/lines.txt
  MNE_HASH_BATCH_INLINE=$v timeout 300 python bench.py --config office0_hash --no-variants --cpu-iters 0 --steps 300 --warmup 50 2>/dev/null | tail -1 | line | tee -a $OUT/lines.txt
done
for k in 1 2; do
  echo -n "office0 default: " | tee -a $OUT/lines.txt
  timeout 300 python bench.py --no-variants --cpu-iters 0 --steps 300 --warmup 50 2>/dev/null | tail -1 | line | tee -a $OUT/lines.txt
  echo -n "office0 driver form: " | tee -a $OUT/lines.txt
  timeout 300 python bench.py --no-variants --cpu-iters 0 --steps 20 --warmup 5 2>/dev/null | tail -1 | line | tee -a $OUT/lines.txt
done
echo -n "render_img: " | tee -a $OUT/lines.txt
timeout 300 python bench.py --mode render_img --cpu-iters 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print({k: d[k] for k in d if k in ('value','ms_per_step','ms_per_pair')}, d.get('roofline',{}).get('frac'))" | tee -a $OUT/lines.txt
