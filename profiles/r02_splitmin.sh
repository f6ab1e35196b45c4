#!/bin/bash
cd "$(dirname "$0")/.."; export TMPDIR=/tmp; mkdir -p gpurun_out
out=gpurun_out/split_min.txt; : > $out
line() { python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('  %-44s ms/step %.4f  it/s %.1f | %s %.3f | %s' % (d['config']['workload'], d['ms_per_step'], d['value'], r['kernel'][:12], r['avg_launch_ms'], ['%s %.3f' % (k[:10], v) for k, v in r['other_kernels_avg_ms'].items() if k[:4] in ('tile', 'ray_')]))"; }
for m in 4096 8192 2048; do echo "MNE_TILE_SPLIT_MIN=$m" >> $out; for c in office0 scannet indoor; do MNE_TILE_SPLIT_MIN=$m timeout 300 python bench.py --config $c --steps 150 --warmup 20 --cpu-iters 0 2>/dev/null | line >> $out; done; done
cat $out
