#!/bin/bash
# critical chain on one stream + device flag hand-off vs event-joined stream roles
cd "$(dirname "$0")/.."; export TMPDIR=/tmp; mkdir -p gpurun_out
out=gpurun_out/one_stream.txt; : > $out
timeout 900 python -m pytest tests/test_hip_parity_gpu.py -q -x -k "full_size or baseline_config or quality or graph or fused_step" 2>&1 | grep -E "passed|failed|error" >> $out
line() { python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('  %-44s ms/step %.4f  it/s %.1f psnr %.2f | %s %.3f | %s' % (d['config']['workload'], d['ms_per_step'], d['value'], d['psnr_last_iter'], r['kernel'][:12], r['avg_launch_ms'], ['%s %.3f' % (k[:10], v) for k, v in r['other_kernels_avg_ms'].items()]))"; }
for v in 1 0 1 0; do echo "MNE_ONE_CRITICAL_STREAM=$v" >> $out; MNE_ONE_CRITICAL_STREAM=$v timeout 300 python bench.py --steps 300 --warmup 30 --cpu-iters 0 2>/dev/null | line >> $out; done
for c in scannet indoor; do for v in 1 0; do echo "MNE_ONE_CRITICAL_STREAM=$v" >> $out; MNE_ONE_CRITICAL_STREAM=$v timeout 300 python bench.py --config $c --steps 150 --warmup 20 --cpu-iters 0 2>/dev/null | line >> $out; done; done
rm -rf /tmp/pf; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pf -o t -- python bench.py --steps 100 --warmup 20 --cpu-iters 0 > /dev/null 2>&1
python profiles/timeline.py $(find /tmp/pf -name '*.db' | head -1) 40 40 >> $out 2>&1
cat $out
