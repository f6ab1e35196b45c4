#!/bin/bash
cd "$(dirname "$0")/.."; export TMPDIR=/tmp; mkdir -p gpurun_out
out=gpurun_out/var2_r02.txt; : > $out
cp mneslam_amd/libmneslam_hip.so /tmp/lib_orig.so
run() {
  for k in 1 2; do env "$@" python bench.py --steps 300 --warmup 30 --cpu-iters 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('  ms/step %.4f  it/s %.1f psnr %.2f | %s %.3f | %s' % (d['ms_per_step'], d['value'], d['psnr_last_iter'], r['kernel'][:12], r['avg_launch_ms'], list(r['other_kernels_avg_ms'].values())))" >> $out; done
}
echo "== default lib, side priority high" >> $out; run MNE_SIDE_PRIORITY=1
echo "== default lib, side priority normal" >> $out; run MNE_SIDE_PRIORITY=0
for lib in profiles/_variants/lib_*.so; do
  cp $lib mneslam_amd/libmneslam_hip.so
  echo "== $lib prio high" >> $out; run MNE_SIDE_PRIORITY=1
  echo "== $lib prio normal" >> $out; run MNE_SIDE_PRIORITY=0
done
cp /tmp/lib_orig.so mneslam_amd/libmneslam_hip.so
cat $out
