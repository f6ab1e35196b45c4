#!/bin/bash
cd "$(dirname "$0")/.."; export TMPDIR=/tmp; mkdir -p gpurun_out
for c in "--config scannet" "--hidden 64" "--config indoor" "--config office0_hash"; do
  echo "== $c"; timeout 300 python bench.py $c --steps 100 --warmup 20 --no-variants --cpu-iters 0 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=d['roofline']
print(d['value'], d['ms_per_step'], r['kernel'][:30], r['avg_launch_ms'], {k[:20]:round(v,4) for k,v in r['other_kernels_avg_ms'].items()})"
done
