#!/bin/bash
# per-XCD list segments (ABI 7): parity tests of the fused paths, then the three plane workloads and the driver's form
cd "$(dirname "$0")/.."; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -rf 2>&1 | grep -E "FAILED|passed|failed" | tail -6
for c in office0 scannet indoor; do for k in 1 2; do
  python bench.py --config $c --steps 200 --warmup 30 --cpu-iters 0 --no-variants 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
o = dict(r['other_kernels_avg_ms']); o[r['kernel']] = r['avg_launch_ms']
print('$c ms/step %.4f it/s %.1f | ' % (d['ms_per_step'], d['value']) + ' '.join('%s=%.3f' % (k.split(' ')[0], v) for k, v in o.items()))"
done; done
for k in 1 2 3; do python bench.py --steps 20 --warmup 5 --cpu-iters 0 --no-variants 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('driver form it/s %.1f' % d['value'])"; done
