#!/bin/bash
# end-of-round GPU pass on the final tree: parity tests, smoke, the default bench line, the driver's form three times
cd "$(dirname "$0")/.."; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -rf 2>&1 | grep -E "FAILED|passed|failed" | tail -6
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/r05_bench_default.json 2> gpurun_out/r05_bench_default.err; tail -c 300 gpurun_out/r05_bench_default.err
: > gpurun_out/r05_bench_driver_form.json
for k in 1 2 3; do timeout 600 python bench.py --steps 20 --warmup 5 --cpu-iters 0 --no-variants 2>/dev/null | tail -1 >> gpurun_out/r05_bench_driver_form.json; done
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r05_bench_driver_form_torchrun.json
cd /tmp; rm -rf /tmp/ks_f; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ks_f -o k -- python $OLDPWD/bench.py --no-variants --cpu-iters 0 --steps 100 --warmup 20 > /dev/null 2>&1
python $OLDPWD/profiles/summarize_rocprof_db.py $(find /tmp/ks_f -name '*.db' | head -1) > $OLDPWD/gpurun_out/r05_kernel_stats_office0_final.txt 2>&1; head -16 $OLDPWD/gpurun_out/r05_kernel_stats_office0_final.txt | cut -c1-150
cd $OLDPWD
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r05_bench_default.json').read().strip().splitlines()[-1])
print('default', d['value'], d['ms_per_step'], d['roofline']['kernel'][:20], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['roofline']['traffic_frac'], d['cpu_baseline']['value'])
for k, v in d['variants'].items():
    print(k, {x: v[x] for x in v if x in ('value', 'ms_per_step', 'ms_per_pair', 'error', 'total_ms')}, v.get('roofline', {}).get('frac', v.get('frac')))
for l in open('gpurun_out/r05_bench_driver_form.json'):
    print('driver form', json.loads(l)['value'])
print('driver form (full line, torchrun)', json.loads(open('gpurun_out/r05_bench_driver_form_torchrun.json').read())['value'])
PY
