#!/bin/bash
cd "$(dirname "$0")/.."; export TMPDIR=/tmp; mkdir -p gpurun_out
out=gpurun_out/ray_lds2.txt; : > $out
timeout 1500 python -m pytest tests/test_hip_parity_gpu.py -q -x 2>&1 | grep -E "passed|failed|error" >> $out
MNE_HOT_LDS_SAMPLES=48 timeout 600 python -m pytest tests/test_hip_parity_gpu.py -q -x -k "full_size or baseline_config" 2>&1 | grep -E "passed|failed|error" >> $out
line() { python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('  %-44s ms/step %.4f  it/s %.1f psnr %.2f | %s %.3f | %s' % (d['config']['workload'], d['ms_per_step'], d['value'], d['psnr_last_iter'], r['kernel'][:12], r['avg_launch_ms'], ['%s %.3f' % (k[:10], v) for k, v in r['other_kernels_avg_ms'].items()]))"; }
for c in office0 indoor; do timeout 300 python bench.py --config $c --steps 150 --warmup 20 --cpu-iters 0 2>/dev/null | line >> $out; done
echo "indoor, no LDS cap:" >> $out; MNE_HOT_LDS_SAMPLES=100000 timeout 300 python bench.py --config indoor --steps 150 --warmup 20 --cpu-iters 0 2>/dev/null | line >> $out
cat $out
