#!/bin/bash
cd "$(dirname "$0")/.."; export TMPDIR=/tmp; mkdir -p gpurun_out
out=gpurun_out/var_r02.txt; : > $out
cp mneslam_amd/libmneslam_hip.so /tmp/lib_orig.so
for lib in /tmp/lib_orig.so profiles/_variants/lib_*.so; do
  [ "$lib" != /tmp/lib_orig.so ] && cp $lib mneslam_amd/libmneslam_hip.so
  echo "== $lib" >> $out
  for k in 1 2; do python bench.py --steps 300 --warmup 30 --cpu-iters 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('  ms/step %.4f  it/s %.1f psnr %.2f | %s %.3f | %s' % (d['ms_per_step'], d['value'], d['psnr_last_iter'], r['kernel'][:12], r['avg_launch_ms'], list(r['other_kernels_avg_ms'].values())))" >> $out; done
  rm -rf /tmp/pv; timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/pv -o t -- python bench.py --steps 60 --warmup 10 --cpu-iters 0 > /dev/null 2>&1
  python profiles/summarize_rocprof_db.py $(find /tmp/pv -name '*.db' | head -1) 70 2>&1 | grep -E "tile_adam_kernel|wgrad_fused|wgrad_reduce" | cut -c1-130 >> $out
done
cp /tmp/lib_orig.so mneslam_amd/libmneslam_hip.so
cat $out
