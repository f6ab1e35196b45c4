#!/bin/bash
# round 5, second GPU pass: full GPU suite on the new tests, default bench line with the new variants
cd "$(dirname "$0")/.."; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
timeout 900 python bench.py > gpurun_out/r05_bench_default.json 2> gpurun_out/r05_bench_default.err; tail -c 600 gpurun_out/r05_bench_default.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r05_bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline'].get('traffic_source'))
for k, v in d['variants'].items():
    print(k, {x: v[x] for x in v if x in ('value', 'ms_per_step', 'ms_per_pair', 'error', 'total_ms', 'decoded_samples_per_pair')}, v.get('roofline', {}).get('frac'))
PY
