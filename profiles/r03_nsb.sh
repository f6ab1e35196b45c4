#!/bin/bash
# NS-b re-measured on the round-3 pipeline: hipGraph replay and fp16 plane storage against eager fp32, same box, alternating
cd "$(dirname "$0")/.."; export TMPDIR=/tmp; mkdir -p gpurun_out
P='import json,sys
d=json.loads(sys.stdin.read().strip().split("\n")[-1]); r=d["roofline"]
print(round(d["value"],1), round(d["ms_per_step"],4), round(r["avg_launch_ms"],4))'
for c in "" "--config indoor"; do
 for rep in 1 2; do
  echo -n "eager fp32 [$c] "; timeout 300 python bench.py $c --steps 200 --warmup 30 --cpu-iters 0 --no-variants | python -c "$P"
  echo -n "graph      [$c] "; MNE_GRAPH=1 timeout 300 python bench.py $c --steps 200 --warmup 30 --cpu-iters 0 --no-variants --event-every 100000 | python -c "$P"
  echo -n "eager, no events [$c] "; timeout 300 python bench.py $c --steps 200 --warmup 30 --cpu-iters 0 --no-variants --event-every 100000 | python -c "$P"
  echo -n "fp16 planes[$c] "; timeout 300 python bench.py $c --plane-storage fp16 --steps 200 --warmup 30 --cpu-iters 0 --no-variants | python -c "$P"
 done
done
