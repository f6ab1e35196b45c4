#!/bin/bash
# round 5, third GPU pass: which GPU test fails; decode_kernel task queue A/B; repeat of the two outlier variants
cd "$(dirname "$0")/.."; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q -rf 2>&1 | grep -E "FAILED|passed|failed|Error|assert" | head -30
for v in noqueue main noqueue main; do
  python profiles/r03_variant_bench.py $v --steps 200 --warmup 30 --cpu-iters 0 --no-variants 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
o = dict(r['other_kernels_avg_ms']); o[r['kernel']] = r['avg_launch_ms']
print('$v ms/step %.4f it/s %.1f | ' % (d['ms_per_step'], d['value']) + ' '.join('%s=%.3f' % (k.split(' ')[0], v) for k, v in o.items()))"
done
for v in noqueue main; do
  python profiles/r03_variant_bench.py $v --steps 20 --warmup 5 --cpu-iters 0 --no-variants 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('driver form $v it/s %.1f' % d['value'])"
done
for c in indoor office0_fp16 indoor office0_fp16; do
  python bench.py --config $c --steps 300 --warmup 30 --cpu-iters 0 --no-variants 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c it/s %.1f' % d['value'])"
done
