#!/bin/bash
cd "$(dirname "$0")/.."; export TMPDIR=/tmp; mkdir -p gpurun_out
for c in "--config scannet" "--config indoor" "" "--hidden 64"; do
  echo "== $c"; timeout 300 python bench.py $c --steps 200 --warmup 20 --no-variants --cpu-iters 0 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=d['roofline']
print(round(d['value'],1), round(d['ms_per_step'],4), r['kernel'][:30], round(r['avg_launch_ms'],4), {k[:20]:round(v,4) for k,v in r['other_kernels_avg_ms'].items()})"
done
timeout 300 python bench.py --steps 20 --warmup 5 --no-variants --cpu-iters 0 | cut -c1-200
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03_gpu_tests.txt 2>&1; grep -E "passed|failed|error" gpurun_out/r03_gpu_tests.txt | tail -5
