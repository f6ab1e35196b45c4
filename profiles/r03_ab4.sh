#!/bin/bash
# same-box A/B with more repetitions: _ab/ vs this tree, alternating, 300 steps, no timing events (pure wall clock)
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
P='import json,sys
d=json.loads(sys.stdin.read().strip().split("\n")[-1]); print(round(d["ms_per_step"]*1000,1), end=" ")'
for t in _ab .; do echo -n "$t: "; for rep in 1 2 3 4 5 6; do :; done; echo; done > /dev/null
for rep in 1 2 3 4 5 6; do
  for t in _ab .; do
    echo -n "$t "; (cd $t; timeout 300 python bench.py --steps 300 --warmup 30 --cpu-iters 0 --event-every 100000 $([ $t = . ] && echo --no-variants) | python -c "$P"); echo
  done
done
