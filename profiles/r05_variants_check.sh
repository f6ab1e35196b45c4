#!/bin/bash
cd "$(dirname "$0")/.."; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python bench.py --cpu-iters 0 > gpurun_out/r05_bench_default2.json 2> gpurun_out/r05_bench_default2.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r05_bench_default2.json').read().strip().splitlines()[-1])
print('default', d['value'])
for k, v in d['variants'].items():
    print(k, {x: v[x] for x in v if x in ('value', 'ms_per_step', 'ms_per_pair', 'error', 'total_ms', 'avg_launch_ms')})
PY
bash profiles/r05_bin_xcd.sh
