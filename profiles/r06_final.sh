#!/bin/bash
# end-of-round GPU pass on the final tree: parity tests, smoke, the default bench line, the driver's form three times + under torchrun
cd "$(dirname "$0")/.."; export TMPDIR=/tmp; mkdir -p gpurun_out/r06_final; O=gpurun_out/r06_final
timeout 1800 python -m pytest tests -m gpu -q -rf 2>&1 | grep -E "FAILED|passed|failed" | tail -6 | tee $O/gpu_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/smoke.txt
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 300 $O/bench_default.err
: > $O/bench_driver_form.json
for k in 1 2 3; do timeout 600 python bench.py --steps 20 --warmup 5 --cpu-iters 0 --no-variants 2>/dev/null | tail -1 >> $O/bench_driver_form.json; done
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_driver_form_torchrun.json
python - <<'PY'
import json
O = 'gpurun_out/r06_final/'
d = json.loads(open(O + 'bench_default.json').read().strip().splitlines()[-1])
print('default', d['value'], d['ms_per_step'], d['roofline']['kernel'][:20], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['roofline']['traffic_frac'], d['cpu_baseline']['value'])
for k, v in d['variants'].items():
    print(k, {x: v[x] for x in v if x in ('value', 'ms_per_step', 'ms_per_pair', 'error', 'total_ms', 'ms_per_iteration')}, v.get('roofline', {}).get('frac', v.get('frac')))
for l in open(O + 'bench_driver_form.json'):
    print('driver form', json.loads(l)['value'])
print('driver form (full line, torchrun)', json.loads(open(O + 'bench_driver_form_torchrun.json').read())['value'])
PY
