#!/bin/bash
# split tile lists (load balance of the plane update): parity + A/B per workload
cd "$(dirname "$0")/.."; export TMPDIR=/tmp; mkdir -p gpurun_out
out=gpurun_out/tile_split.txt; : > $out
timeout 1500 python -m pytest tests/test_hip_parity_gpu.py -q -x 2>&1 | tail -4
MNE_TILE_SPLIT_MIN=64 timeout 900 python -m pytest tests/test_hip_parity_gpu.py -q -x -k "full_size or baseline_config or mapping_iterations or overflow" 2>&1 | tail -3
line() { python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('  %-52s ms/step %.4f  it/s %.1f psnr %.2f | %s %.3f | %s' % (d['config']['workload'], d['ms_per_step'], d['value'], d['psnr_last_iter'], r['kernel'][:12], r['avg_launch_ms'], ['%s %.3f' % (k[:10], v) for k, v in r['other_kernels_avg_ms'].items()]))"; }
for c in "office0" "office0 --hidden 64" apartment scannet "scannet --hidden 64" indoor; do for sp in 0 1; do
  echo "MNE_NO_TILE_SPLIT=$sp" >> $out
  MNE_NO_TILE_SPLIT=$sp timeout 300 python bench.py --config $c --steps 150 --warmup 20 --cpu-iters 0 2>/dev/null | line >> $out
done; done
cat $out
