#!/bin/bash
cd "$(dirname "$0")/.."; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
out=gpurun_out/hot_r02.txt; : > $out
cp mneslam_amd/libmneslam_hip.so /tmp/lib_orig.so
for lib in /tmp/lib_orig.so profiles/_variants/lib_wpb8.so; do
  [ "$lib" != /tmp/lib_orig.so ] && cp $lib mneslam_amd/libmneslam_hip.so
  echo "== $lib" >> $out
  python bench.py --steps 200 --warmup 20 --cpu-iters 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  ms/step %.4f  it/s %.1f psnr %.2f' % (d['ms_per_step'], d['value'], d['psnr_last_iter']))" >> $out
  rm -rf /tmp/pv; timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/pv -o t -- python bench.py --steps 60 --warmup 10 --cpu-iters 0 > /dev/null 2>&1
  db=$(find /tmp/pv -name '*.db' | head -1)
  python profiles/summarize_rocprof_db.py $db 70 2>&1 | grep -E "ray_kernel|ray_composite|backward_tile|decode_kernel|gather_kernel|tile_adam_kernel|wgrad_fused" | cut -c1-130 >> $out
  python profiles/timeline.py $db 30 20 | head -14 >> $out 2>&1
done
cp /tmp/lib_orig.so mneslam_amd/libmneslam_hip.so
cat $out
