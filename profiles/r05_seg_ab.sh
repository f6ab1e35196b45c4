#!/bin/bash
# tile-list segments: 8 (main: one per XCD) vs 16 / 32 (several per XCD), same box
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
for c in office0 scannet indoor; do for v in main seg16 seg32 main seg16; do
  python profiles/r03_variant_bench.py $v --config $c --steps 200 --warmup 30 --cpu-iters 0 --no-variants 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
o = dict(r['other_kernels_avg_ms']); o[r['kernel']] = r['avg_launch_ms']
print('$c $v ms/step %.4f it/s %.1f | ' % (d['ms_per_step'], d['value']) + ' '.join('%s=%.3f' % (k.split(' ')[0], v) for k, v in o.items()))"
done; done
