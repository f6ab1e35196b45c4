#!/bin/bash
# default bench line (variants + same-batch CPU baseline), driver form, then the counter passes
cd "$(dirname "$0")/.."; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python bench.py > gpurun_out/r03_bench_default.json 2> gpurun_out/r03_bench_default.err; tail -3 gpurun_out/r03_bench_default.err; cut -c1-600 gpurun_out/r03_bench_default.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03_bench_default.json').read().strip().split('\n')[-1])
print(d['value'], d['ms_per_step']); print(json.dumps(d.get('cpu_baseline'))); print(json.dumps(d.get('variants'), indent=1))
PY
timeout 300 python bench.py --steps 20 --warmup 5 --no-variants --cpu-iters 0 | cut -c1-300
bash profiles/r03_pmc.sh
cat gpurun_out/r03_pmc_traffic.txt | cut -c1-170
