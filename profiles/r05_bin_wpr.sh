#!/bin/bash
# bin_kernel: 1 / 2 / 4 waves per ray (A/B), office0 + scannet + indoor
cd "$(dirname "$0")/.."; export TMPDIR=/tmp; mkdir -p gpurun_out
for c in office0 scannet indoor; do for v in binwpr1 binwpr2 main binwpr1 main; do
  python profiles/r03_variant_bench.py $v --config $c --steps 200 --warmup 30 --cpu-iters 0 --no-variants 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
o = dict(r['other_kernels_avg_ms']); o[r['kernel']] = r['avg_launch_ms']
print('$c $v ms/step %.4f it/s %.1f | ' % (d['ms_per_step'], d['value']) + ' '.join('%s=%.3f' % (k.split(' ')[0], v) for k, v in o.items()))"
done; done
python profiles/r03_variant_bench.py main --steps 20 --warmup 5 --cpu-iters 0 --no-variants 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('driver form main it/s %.1f' % d['value'])"
