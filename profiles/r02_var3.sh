#!/bin/bash
# tile_adam variants (operand prefetch behind the first pass, empty-tile fast path, 3 workgroups per CU): parity subset + bench
cd "$(dirname "$0")/.."; export TMPDIR=/tmp; mkdir -p gpurun_out
out=gpurun_out/var3_r02.txt; : > $out
cp mneslam_amd/libmneslam_hip.so /tmp/lib_orig.so
run() {
  for k in 1 2; do python bench.py --steps 300 --warmup 30 --cpu-iters 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('  ms/step %.4f  it/s %.1f psnr %.2f | %s %.3f | %s' % (d['ms_per_step'], d['value'], d['psnr_last_iter'], r['kernel'][:12], r['avg_launch_ms'], ['%s %.3f' % (k[:10], v) for k, v in r['other_kernels_avg_ms'].items()]))" >> $out; done
}
echo "== default lib (empty-tile fast path)" >> $out; run
timeout 300 python bench.py --steps 50 --warmup 10 --cpu-iters 0 2>&1 | tail -1 | cut -c 1-2500
for lib in profiles/_variants/lib_*.so; do
  cp $lib mneslam_amd/libmneslam_hip.so
  echo "== $lib" >> $out
  timeout 300 python -m pytest tests/test_hip_parity_gpu.py -q -x -k "fused_step or full_size or tile_adam or binned" 2>&1 | tail -1 >> $out
  run
done
cp /tmp/lib_orig.so mneslam_amd/libmneslam_hip.so
cat $out
