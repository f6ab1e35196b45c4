#!/bin/bash
# DRAM-locality experiments on tile_adam_kernel: Adam moments stored tile-major (private padded buffers), spatial tile order
cd "$(dirname "$0")/.."; export TMPDIR=/tmp; mkdir -p gpurun_out
out=gpurun_out/var4_r02.txt; : > $out
cp mneslam_amd/libmneslam_hip.so /tmp/lib_orig.so
run() {
  for k in 1 2; do env "$@" python bench.py --steps 300 --warmup 30 --cpu-iters 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('  ms/step %.4f  it/s %.1f psnr %.2f | %s %.3f | %s' % (d['ms_per_step'], d['value'], d['psnr_last_iter'], r['kernel'][:12], r['avg_launch_ms'], ['%s %.3f' % (k[:10], v) for k, v in r['other_kernels_avg_ms'].items()]))" >> $out; done
}
echo "== default lib" >> $out; run A=1
cp profiles/_variants/lib_ident.so mneslam_amd/libmneslam_hip.so; echo "== spatial order" >> $out; run A=1
cp profiles/_variants/lib_mvtile.so mneslam_amd/libmneslam_hip.so; echo "== m, v tile-major" >> $out; run MNE_EXP_TILEMAJOR=1
cp profiles/_variants/lib_mvtile_ident.so mneslam_amd/libmneslam_hip.so; echo "== m, v tile-major + spatial order" >> $out; run MNE_EXP_TILEMAJOR=1
cp /tmp/lib_orig.so mneslam_amd/libmneslam_hip.so
cat $out
