#!/bin/bash
# the fused form on the training path once more, final tree: the a-priori tiles gather INLINE in decode_kernel (variant inlineg, -DMNE_INLINE_GATHER_EXPERIMENT: no gather_kernel) vs the split form
cd "$(dirname "$0")/.."; REPO=$PWD; export TMPDIR=/tmp; OUT=$REPO/gpurun_out/r06_inline_gather; mkdir -p $OUT
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value'],1), 'it/s', round(d['ms_per_step'],4), 'ms;', r['kernel'][:22], round(r['avg_launch_ms']*1000,1), 'us', {k[:14]: round(v*1000,1) for k,v in r['other_kernels_avg_ms'].items()})"; }
for cfg in ${CFGS:-office0 scannet indoor}; do
for v in ${VARS:-main inlineg main inlineg}; do
  echo -n "$cfg $v: " | tee -a $OUT/lines.txt
  timeout 300 python profiles/r03_variant_bench.py $v --config $cfg --no-variants --cpu-iters 0 --steps 300 --warmup 50 2>/dev/null | tail -1 | line | tee -a $OUT/lines.txt
done; done
for v in ${VARS:-main inlineg}; do for k in 1 2; do
  echo -n "office0 driver form $v: " | tee -a $OUT/lines.txt
  timeout 300 python profiles/r03_variant_bench.py $v --no-variants --cpu-iters 0 --steps 20 --warmup 5 2>/dev/null | tail -1 | line | tee -a $OUT/lines.txt
done; done
