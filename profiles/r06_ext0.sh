#!/bin/bash
# decode_kernel's tail: is it the resolver's extension tile (serial in one wave, inline gather)?  variant ext0 = no extension (unresolved rays go to the deferred pass)
cd "$(dirname "$0")/.."; REPO=$PWD; export TMPDIR=/tmp; OUT=$REPO/gpurun_out/r06_ext0; mkdir -p $OUT
for cfg in office0 scannet indoor; do
for v in main ext0 main ext0; do
  echo -n "$cfg $v: " | tee -a $OUT/lines.txt
  timeout 300 python profiles/r03_variant_bench.py $v --config $cfg --no-variants --cpu-iters 0 --steps 300 --warmup 50 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value'],1), 'it/s', round(d['ms_per_step'],4), 'ms; tile_adam', round(r['avg_launch_ms']*1000,1), 'us', {k[:14]: round(v*1000,1) for k,v in r['other_kernels_avg_ms'].items()})" | tee -a $OUT/lines.txt
done; done
