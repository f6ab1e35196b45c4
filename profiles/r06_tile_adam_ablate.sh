#!/bin/bash
# tile_adam_kernel on the list-bound workloads: where does a pass spend its time?  skipd = step D (walk of the sorted contributions) compiled out,
# skipa = no entry is fetched at all (the passes keep their barriers and LDS clears).  Timing only -- the results of these builds are wrong by construction.
cd "$(dirname "$0")/.."; REPO=$PWD; export TMPDIR=/tmp; OUT=$REPO/gpurun_out/r06_tile_adam_ablate; mkdir -p $OUT
for cfg in indoor scannet office0; do
for v in main skipd skipa; do
  echo -n "$cfg $v: " | tee -a $OUT/lines.txt
  timeout 300 python profiles/r03_variant_bench.py $v --config $cfg --no-variants --cpu-iters 0 --steps 200 --warmup 30 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; o=r.get('other_kernels_avg_ms',{}); print(round(d['value'],1), 'it/s', round(d['ms_per_step'],4), 'ms;', r['kernel'][:24], round(r['avg_launch_ms']*1000,1), 'us;', {k[:12]: round(v*1000,1) for k,v in o.items() if 'tile_adam' in k or 'adam' in k})" | tee -a $OUT/lines.txt
done; done
