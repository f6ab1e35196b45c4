#!/bin/bash
# adaptive a-priori prefix (fresh map): thresholds for entering / leaving the decode-everything state (fractions 1 / n of the rays), driver form
cd "$(dirname "$0")/.."; REPO=$PWD; export TMPDIR=/tmp; OUT=$REPO/gpurun_out/r06_adapt; mkdir -p $OUT
for rep in 1 2 3; do
for v in "8 16" "8 8" "8 4" "4 8" "4 4" "2 4" "2 2" "16 16"; do
  set -- $v
  echo -n "enter 1/$1 leave 1/$2: " | tee -a $OUT/lines.txt
  MNE_ADAPT_ENTER=$1 MNE_ADAPT_LEAVE=$2 timeout 120 python bench.py --no-variants --cpu-iters 0 --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'it/s', round(d['ms_per_step'],4), 'ms')" | tee -a $OUT/lines.txt
done; done
echo -n "MNE_NO_ADAPT=1: " | tee -a $OUT/lines.txt
MNE_NO_ADAPT=1 timeout 120 python bench.py --no-variants --cpu-iters 0 --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'it/s', round(d['ms_per_step'],4), 'ms')" | tee -a $OUT/lines.txt
for v in "8 16" "4 4" "2 2"; do set -- $v; echo "per-step times, enter 1/$1 leave 1/$2:" | tee -a $OUT/lines.txt; MNE_ADAPT_ENTER=$1 MNE_ADAPT_LEAVE=$2 python profiles/r03_step_times.py 40 2>/dev/null | head -1 | cut -c1-260 | tee -a $OUT/lines.txt; done
