#!/bin/bash
cd "$(dirname "$0")/.."; export TMPDIR=/tmp; mkdir -p gpurun_out
out=gpurun_out/hash_dense_wgs.txt; : > $out
cp mneslam_amd/libmneslam_hip.so /tmp/lib_orig.so
run() { echo "== $1" >> $out; for k in 1 2; do timeout 300 python bench.py --config office0_hash --steps 200 --warmup 20 --cpu-iters 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('  ms/step %.4f  it/s %.1f psnr %.2f | %s %.3f' % (d['ms_per_step'], d['value'], d['psnr_last_iter'], r['kernel'][:12], r['avg_launch_ms']))" >> $out; done; }
run "default (64 workgroups per dense level)"
for lib in profiles/_variants/lib_w*.so; do cp $lib mneslam_amd/libmneslam_hip.so; run $lib; done
cp /tmp/lib_orig.so mneslam_amd/libmneslam_hip.so
cat $out
