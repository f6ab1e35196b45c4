#!/bin/bash
# wave-major (the tree) vs workgroup-major (variant wgmajor, -DMNE_WAVE_MAJOR=0) task numbering, same box, alternating
cd "$(dirname "$0")/.."; REPO=$PWD; export TMPDIR=/tmp; OUT=$REPO/gpurun_out/r06_wave_major; mkdir -p $OUT
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value'],1), 'it/s', round(d['ms_per_step'],4), 'ms;', r['kernel'][:22], round(r['avg_launch_ms']*1000,1), 'us', {k[:14]: round(v*1000,1) for k,v in r['other_kernels_avg_ms'].items()})"; }
for cfg in ${CFGS:-indoor office0 scannet apartment office0_hash}; do
for v in ${VARS:-wgmajor main wgmajor main}; do
  echo -n "$cfg $v: " | tee -a $OUT/ab.txt
  timeout 300 python profiles/r03_variant_bench.py $v --config $cfg --no-variants --cpu-iters 0 --steps 300 --warmup 50 2>/dev/null | tail -1 | line | tee -a $OUT/ab.txt
done; done
