#!/bin/bash
# alternating stream roles of the fused step (MNE_PINGPONG=1: the next render follows the plane update in the same queue) vs fixed roles, same box, alternating
cd "$(dirname "$0")/.."; REPO=$PWD; export TMPDIR=/tmp; OUT=$REPO/gpurun_out/r06_pingpong; mkdir -p $OUT
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value'],1), 'it/s', round(d['ms_per_step'],4), 'ms; tile_adam', round(r['avg_launch_ms']*1000,1), 'us')"; }
for cfg in office0 scannet indoor; do
for rep in 1 2; do
for v in 0 1; do
  echo -n "$cfg 300 steps MNE_PINGPONG=$v: " | tee -a $OUT/lines.txt
  MNE_PINGPONG=$v timeout 300 python bench.py --config $cfg --no-variants --cpu-iters 0 --steps 300 --warmup 50 2>/dev/null | tail -1 | line | tee -a $OUT/lines.txt
done; done; done
for rep in 1 2 3; do
for v in 0 1; do
  echo -n "office0 driver form MNE_PINGPONG=$v: " | tee -a $OUT/lines.txt
  MNE_PINGPONG=$v timeout 300 python bench.py --no-variants --cpu-iters 0 --steps 20 --warmup 5 2>/dev/null | tail -1 | line | tee -a $OUT/lines.txt
done; done
