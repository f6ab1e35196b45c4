#!/bin/bash
# round 5, fourth GPU pass: apartment overlap test traceback; balanced decode schedule A/B; new tests
cd "$(dirname "$0")/.."; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_parity_gpu.py -m gpu -q -x -k "apartment" 2>&1 | grep -v "^$" | tail -40
timeout 900 python -m pytest tests/test_hip_parity_gpu.py -m gpu -q -k "hash or pose or fp16_planes_autograd or bench_path or fused" 2>&1 | tail -5
for v in unbalanced main unbalanced main; do
  python profiles/r03_variant_bench.py $v --steps 200 --warmup 30 --cpu-iters 0 --no-variants 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
o = dict(r['other_kernels_avg_ms']); o[r['kernel']] = r['avg_launch_ms']
print('$v ms/step %.4f it/s %.1f | ' % (d['ms_per_step'], d['value']) + ' '.join('%s=%.3f' % (k.split(' ')[0], v) for k, v in o.items()))"
done
for v in unbalanced main; do
  python profiles/r03_variant_bench.py $v --steps 20 --warmup 5 --cpu-iters 0 --no-variants 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('driver form $v it/s %.1f' % d['value'])"
done
for c in indoor scannet office0_hash; do for v in unbalanced main; do
  python profiles/r03_variant_bench.py $v --config $c --steps 300 --warmup 30 --cpu-iters 0 --no-variants 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c $v it/s %.1f' % d['value'])"
done; done
