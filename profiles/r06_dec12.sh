#!/bin/bash
# decode_kernel at 12 waves per workgroup (3 per SIMD, 168 registers: 8 spilled in the office0 instantiation) vs 8 (the tree), now that the balanced schedule gives every wave the same tile count
cd "$(dirname "$0")/.."; REPO=$PWD; export TMPDIR=/tmp; OUT=$REPO/gpurun_out/r06_dec12; mkdir -p $OUT
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value'],1), 'it/s', round(d['ms_per_step'],4), 'ms;', r['kernel'][:22], round(r['avg_launch_ms']*1000,1), 'us', {k[:14]: round(v*1000,1) for k,v in r['other_kernels_avg_ms'].items()})"; }
for cfg in office0 apartment; do
for v in main dec12 main dec12; do
  echo -n "$cfg $v: " | tee -a $OUT/lines.txt
  timeout 300 python profiles/r03_variant_bench.py $v --config $cfg --no-variants --cpu-iters 0 --steps 300 --warmup 50 2>/dev/null | tail -1 | line | tee -a $OUT/lines.txt
done; done
for v in main dec12; do
  echo -n "office0 driver form $v: " | tee -a $OUT/lines.txt
  timeout 300 python profiles/r03_variant_bench.py $v --no-variants --cpu-iters 0 --steps 20 --warmup 5 2>/dev/null | tail -1 | line | tee -a $OUT/lines.txt
done
