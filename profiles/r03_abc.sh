#!/bin/bash
# three-way same-box comparison, alternating, 300 steps, no timing events: _ab/ tree, this tree, a variant build of this tree
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
P='import json,sys
d=json.loads(sys.stdin.read().strip().split("\n")[-1]); print(round(d["ms_per_step"]*1000,1), end=" ")'
V=${1:-x_noov}
for rep in 1 2 3 4 5; do
  echo -n "_ab "; (cd _ab; timeout 300 python bench.py --steps 300 --warmup 30 --cpu-iters 0 --event-every 100000 | python -c "$P"); echo
  echo -n "cur "; timeout 300 python bench.py --steps 300 --warmup 30 --cpu-iters 0 --event-every 100000 --no-variants | python -c "$P"; echo
  echo -n "$V "; timeout 300 python profiles/r03_variant_bench.py $V --steps 300 --warmup 30 --cpu-iters 0 --event-every 100000 --no-variants | python -c "$P"; echo
done
