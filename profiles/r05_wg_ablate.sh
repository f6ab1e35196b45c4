#!/bin/bash
# in-kernel decoder weight gradients (ray_kernel<..., WG>): what bounds it -- the B-operand loads, the MFMAs, or the rest
cd "$(dirname "$0")/.."; export TMPDIR=/tmp; mkdir -p gpurun_out
for v in main wg_noloads wg_nomfma; do
  python profiles/r03_variant_bench.py $v --steps 200 --warmup 30 --cpu-iters 0 --no-variants 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
o = dict(r['other_kernels_avg_ms']); o[r['kernel']] = r['avg_launch_ms']
print('$v ms/step %.4f it/s %.1f | ' % (d['ms_per_step'], d['value']) + ' '.join('%s=%.3f' % (k.split(' ')[0], v) for k, v in o.items()))"
done
MNE_WGRAD_INLINE=0 python profiles/r03_variant_bench.py main --steps 200 --warmup 30 --cpu-iters 0 --no-variants 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
o = dict(r['other_kernels_avg_ms']); o[r['kernel']] = r['avg_launch_ms']
print('second-pass ms/step %.4f it/s %.1f | ' % (d['ms_per_step'], d['value']) + ' '.join('%s=%.3f' % (k.split(' ')[0], v) for k, v in o.items()))"
