#!/bin/bash
# wave-major task numbering in decode_kernel / ray_kernel<...,4> / heavy_bwd_kernel + ray grid over all CUs: the tree, every workload
cd "$(dirname "$0")/.."; REPO=$PWD; export TMPDIR=/tmp; OUT=$REPO/gpurun_out/r06_wave_major; mkdir -p $OUT
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value'],1), 'it/s', round(d['ms_per_step'],4), 'ms;', r['kernel'][:22], round(r['avg_launch_ms']*1000,1), 'us', {k[:14]: round(v*1000,1) for k,v in r['other_kernels_avg_ms'].items()})"; }
for cfg in office0 apartment scannet indoor office0_hash; do
for k in 1 2; do
  echo -n "$cfg: " | tee -a $OUT/lines.txt
  timeout 300 python bench.py --config $cfg --no-variants --cpu-iters 0 --steps 300 --warmup 50 2>/dev/null | tail -1 | line | tee -a $OUT/lines.txt
done; done
for k in 1 2 3; do
  echo -n "office0 driver form: " | tee -a $OUT/lines.txt
  timeout 300 python bench.py --no-variants --cpu-iters 0 --steps 20 --warmup 5 2>/dev/null | tail -1 | line | tee -a $OUT/lines.txt
done
