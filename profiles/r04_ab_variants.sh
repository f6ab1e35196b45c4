#!/bin/bash
# same-box A/B of library variants (mneslam_amd/_fuzz/<name>/, "main" = the shipped build): usage  r04_ab_variants.sh "<bench args>" reps v1 v2 ...
cd "$(dirname "$0")/.."; export TMPDIR=/tmp; mkdir -p gpurun_out
ARGS="$1"; REPS=$2; shift 2
P='import json,sys
d=json.loads(sys.stdin.read().strip().split("\n")[-1]); r=d["roofline"]
print(round(d["value"],1), round(d["ms_per_step"],4), r["kernel"][:24], round(r["avg_launch_ms"],4), {k[:14]: round(v,3) for k,v in r.get("other_kernels_avg_ms",{}).items()})'
for rep in $(seq $REPS); do
  for v in "$@"; do
    echo -n "$v   "; timeout 300 python profiles/r03_variant_bench.py $v $ARGS --cpu-iters 0 --no-variants 2>/dev/null | python -c "$P"
  done
done
