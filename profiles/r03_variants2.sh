#!/bin/bash
# tile_adam_kernel occupancy variants against the shipped build, same box, alternating (office0, 200 steps)
cd "$(dirname "$0")/.."; export TMPDIR=/tmp; mkdir -p gpurun_out
P='import json,sys
d=json.loads(sys.stdin.read().strip().split("\n")[-1]); r=d["roofline"]
print(round(d["value"],1), round(d["ms_per_step"],4), round(r["avg_launch_ms"],4))'
for rep in 1 2; do
  for v in "$@"; do
    echo -n "$v  "; timeout 300 python profiles/r03_variant_bench.py $v --steps 200 --warmup 20 --cpu-iters 0 --no-variants | python -c "$P"
  done
done
