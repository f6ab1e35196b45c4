#!/bin/bash
# the tile-parallel backward (heavy_bwd_kernel, INS Indoor's path) on SHORT-ray batches: rays with more than T backward tiles only composite in ray_kernel,
# their tiles are walked tile-parallel afterwards.  MNE_HEAVY_NTILE=1 enables the heavy list on 4-tile rays, MNE_HEAVY_TILES=T.
cd "$(dirname "$0")/.."; REPO=$PWD; export TMPDIR=/tmp; OUT=$REPO/gpurun_out/r06_heavy_all; mkdir -p $OUT
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value'],1), 'it/s', round(d['ms_per_step'],4), 'ms;', r['kernel'][:22], round(r['avg_launch_ms']*1000,1), 'us', {k[:14]: round(v*1000,1) for k,v in r['other_kernels_avg_ms'].items()})"; }
for cfg in office0 scannet apartment; do
for v in off 1 2 off 1 2; do
  echo -n "$cfg heavy_tiles=$v: " | tee -a $OUT/lines.txt
  if [ $v = off ]; then E=""; else E="MNE_HEAVY_NTILE=1 MNE_HEAVY_TILES=$v"; fi
  env $E timeout 300 python bench.py --config $cfg --no-variants --cpu-iters 0 --steps 300 --warmup 50 2>/dev/null | tail -1 | line | tee -a $OUT/lines.txt
done; done
