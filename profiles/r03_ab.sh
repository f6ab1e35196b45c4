#!/bin/bash
# same-box A/B: the tree under _ab/ (a worktree of an earlier commit, built there) against this tree, alternating
cd "$(dirname "$0")/.."; export TMPDIR=/tmp; mkdir -p gpurun_out
P='import json,sys
d=json.loads(sys.stdin.read().strip().split("\n")[-1]); r=d["roofline"]
print(round(d["value"],1), round(d["ms_per_step"],4), round(r["avg_launch_ms"],4), {k[:14]:round(v,4) for k,v in r["other_kernels_avg_ms"].items()})'
for c in "${@:-}"; do
 for rep in 1 2; do
  for t in _ab .; do
    echo -n "$t [$c] "; (cd $t; timeout 300 python bench.py $c --steps 200 --warmup 20 --cpu-iters 0 $([ $t = . ] && echo --no-variants) | python -c "$P")
  done
 done
done
