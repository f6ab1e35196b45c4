#!/bin/bash
# tile_adam_kernel: nontemporal cache policy for the Adam sweep's streams (MNE_ADAM_NT bit mask: 1 m/v stores, 2 m/v loads, 4 p loads, 8 p stores)
cd "$(dirname "$0")/.."; REPO=$PWD; export TMPDIR=/tmp; OUT=$REPO/gpurun_out/r05_adam_nt; mkdir -p $OUT
for cfg in office0 scannet indoor; do
for v in main nt1 nt3 nt11 nt15 main; do
  echo -n "$cfg $v: " | tee -a $OUT/lines.txt
  timeout 300 python profiles/r03_variant_bench.py $v --config $cfg --no-variants --cpu-iters 0 --steps 300 --warmup 50 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value'],1), 'it/s', round(d['ms_per_step'],4), 'ms; tile_adam', round(r['avg_launch_ms']*1000,1), 'us')" | tee -a $OUT/lines.txt
done; done
