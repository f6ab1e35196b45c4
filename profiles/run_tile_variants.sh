#!/bin/bash
# On the GPU box: time the fused iteration with each tile_adam variant.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; rm -f gpurun_out/tile_variants.log
cp mneslam_amd/libmneslam_hip.so /tmp/lib_orig.so
for lib in profiles/_variants/lib_*.so; do
  cp $lib mneslam_amd/libmneslam_hip.so
  echo "== $lib" >> gpurun_out/tile_variants.log
  python bench.py --steps 60 --warmup 10 --cpu-iters 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']
print('  ms/step %.3f  dom=%s %.3f ms  others=%s' % (d['ms_per_step'], r['kernel'][:14], r['avg_launch_ms'], r['other_kernels_avg_ms']))
" >> gpurun_out/tile_variants.log
done
cp /tmp/lib_orig.so mneslam_amd/libmneslam_hip.so
cat gpurun_out/tile_variants.log
