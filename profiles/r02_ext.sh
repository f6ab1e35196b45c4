#!/bin/bash
# resolver extension cap (tiles decoded serially beyond the a-priori prefix before deferring the ray): A/B per workload
cd "$(dirname "$0")/.."; export TMPDIR=/tmp; mkdir -p gpurun_out
out=gpurun_out/resolver_ext.txt; : > $out
cp mneslam_amd/libmneslam_hip.so /tmp/lib_orig.so
line() { python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('  %-44s ms/step %.4f  it/s %.1f psnr %.2f | %s %.3f | %s' % (d['config']['workload'], d['ms_per_step'], d['value'], d['psnr_last_iter'], r['kernel'][:12], r['avg_launch_ms'], ['%s %.3f' % (k[:10], v) for k, v in r['other_kernels_avg_ms'].items()]))"; }
run() { for c in office0 scannet indoor; do timeout 300 python bench.py --config $c --steps 150 --warmup 20 --cpu-iters 0 2>/dev/null | line >> $out; done; }
echo "== default (cap 2)" >> $out; run
timeout 600 python -m pytest tests/test_hip_parity_gpu.py -q -x -k "full_size or baseline_config" 2>&1 | tail -2 >> $out
for lib in profiles/_variants/lib_ext*.so; do cp $lib mneslam_amd/libmneslam_hip.so; echo "== $lib" >> $out; run; done
cp /tmp/lib_orig.so mneslam_amd/libmneslam_hip.so
cat $out
