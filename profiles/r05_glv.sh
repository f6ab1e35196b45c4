#!/bin/bash
# inline gathers of the tile kernels with 12 (glv1) or 24 (main) corner rows in flight: A/B
cd "$(dirname "$0")/.."; export TMPDIR=/tmp; mkdir -p gpurun_out
for c in office0 scannet indoor; do for v in glv1 main glv1 main; do
  python profiles/r03_variant_bench.py $v --config $c --steps 200 --warmup 30 --cpu-iters 0 --no-variants 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
o = dict(r['other_kernels_avg_ms']); o[r['kernel']] = r['avg_launch_ms']
print('$c $v ms/step %.4f it/s %.1f | ' % (d['ms_per_step'], d['value']) + ' '.join('%s=%.3f' % (k.split(' ')[0], v) for k, v in o.items()))"
done; done
for v in glv1 main glv1 main; do
  python profiles/r03_variant_bench.py $v --steps 20 --warmup 5 --cpu-iters 0 --no-variants 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('driver form $v it/s %.1f' % d['value'])"
done
for v in glv1 main; do
  python profiles/r03_variant_bench.py $v --mode render_img --pretrain 100 --steps 100 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('render_img $v ms/pair %.2f frac %.3f depth_l1 %.5f' % (d['ms_per_step'], d['roofline']['frac'], d['config']['depth_l1_vs_gt']))"
done
