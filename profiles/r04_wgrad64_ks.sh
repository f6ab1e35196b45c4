#!/bin/bash
# wgrad_fused64_kernel: row pairs per batch (WG_KS64 = 2 shipped so far / 4 / 8) and partial slots (256 / 512), same box, alternating;
# workloads: the hash-grid iteration and the tri-plane iteration with the 2x64 decoders (office0, 2150 x 128)
cd "$(dirname "$0")/.."; export TMPDIR=/tmp; mkdir -p gpurun_out
P='import json,sys
d=json.loads(sys.stdin.read().strip().split("\n")[-1]); r=d["roofline"]
print(round(d["value"],1), round(d["ms_per_step"],4), r["kernel"][:28], round(r["avg_launch_ms"],4))'
for rep in 1 2; do
  for v in "$@"; do
    echo -n "$v hash   "; timeout 300 python profiles/r03_variant_bench.py $v --config office0_hash --steps 200 --warmup 20 --cpu-iters 0 --no-variants | python -c "$P"
    echo -n "$v 2x64   "; timeout 300 python profiles/r03_variant_bench.py $v --config office0 --hidden 64 --steps 200 --warmup 20 --cpu-iters 0 --no-variants | python -c "$P"
  done
done
