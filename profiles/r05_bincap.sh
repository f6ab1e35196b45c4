#!/bin/bash
# grid cap of bin_kernel (workgroups per CU): 8 (all wave slots) / 4 / 2 (main) / 1
cd "$(dirname "$0")/.."; export TMPDIR=/tmp; mkdir -p gpurun_out
for c in indoor office0 scannet; do for v in bincap8 bincap4 main bincap1 bincap8 main; do
  python profiles/r03_variant_bench.py $v --config $c --steps 200 --warmup 30 --cpu-iters 0 --no-variants 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
o = dict(r['other_kernels_avg_ms']); o[r['kernel']] = r['avg_launch_ms']
print('$c $v ms/step %.4f it/s %.1f | ' % (d['ms_per_step'], d['value']) + ' '.join('%s=%.3f' % (k.split(' ')[0], v) for k, v in o.items()))"
done; done
