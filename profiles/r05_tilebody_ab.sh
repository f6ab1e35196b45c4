#!/bin/bash
# ray_kernel's training tile body as the shared function hot_backward_tile (main) vs inline (prevtile): same box
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_parity_gpu.py tests/test_layout_fuzz_gpu.py -m gpu -q -k "fused or bench_path or baseline_config or mapping or layout or fuzz" 2>&1 | tail -3
for c in "office0" "scannet" "office0 --hidden 64"; do for v in prevtile main prevtile main; do
  python profiles/r03_variant_bench.py $v --config $c --steps 200 --warmup 30 --cpu-iters 0 --no-variants 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
o = dict(r['other_kernels_avg_ms']); o[r['kernel']] = r['avg_launch_ms']
print('$c $v ms/step %.4f it/s %.1f | ' % (d['ms_per_step'], d['value']) + ' '.join('%s=%.3f' % (k.split(' ')[0], v) for k, v in o.items() if 'ray' in k))"
done; done
