#!/bin/bash
# where the decoder's weight-gradient chain of the hash-grid iteration starts: beside the row sort / beside the slice + Adam launch (shipped) /
# behind the whole table update (then beside the next gather); same box, alternating
cd "$(dirname "$0")/.."; export TMPDIR=/tmp; mkdir -p gpurun_out
P='import json,sys
d=json.loads(sys.stdin.read().strip().split("\n")[-1]); r=d["roofline"]
print(round(d["value"],1), round(d["ms_per_step"],4), r["kernel"][:28], round(r["avg_launch_ms"],4), {k[:14]: round(v,3) for k,v in r.get("other_kernels_avg_ms",{}).items()})'
for rep in 1 2; do
  for v in slice end bin; do
    echo -n "$v   "; MNE_HASH_WGRAD_AT=$v timeout 300 python bench.py --config office0_hash --steps 300 --warmup 30 --cpu-iters 0 --no-variants | python -c "$P"
  done
done
