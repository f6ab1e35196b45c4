#!/bin/bash
# timing ablations of decode_kernel / ray_kernel (results are WRONG in these builds; timing only)
cd "$(dirname "$0")/.."; export TMPDIR=/tmp; mkdir -p gpurun_out; out=gpurun_out/abl_r02.txt; : > $out
cp mneslam_amd/libmneslam_hip.so /tmp/lib_orig.so
for lib in /tmp/lib_orig.so profiles/_variants/lib_*.so; do
  [ "$lib" != /tmp/lib_orig.so ] && cp $lib mneslam_amd/libmneslam_hip.so
  echo "== $lib" >> $out
  rm -rf /tmp/pv; timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/pv -o t -- python bench.py --steps 60 --warmup 10 --cpu-iters 0 > /dev/null 2>&1
  python profiles/summarize_rocprof_db.py $(find /tmp/pv -name '*.db' | head -1) 70 2>&1 | grep -E "ray_kernel|decode_kernel|tile_adam_kernel|wgrad_fused" | cut -c1-130 >> $out
done
cp /tmp/lib_orig.so mneslam_amd/libmneslam_hip.so
cat $out
bash profiles/pmc_sq.sh "_kernel" > /dev/null 2>&1
python - <<PY
import re
txt=open("gpurun_out/pmc_sq.txt").read().split("\n")
show=False
for l in txt:
    if not l.startswith(" "):
        show = ("decode_kernel" in l) or ("ray_kernel" in l) or ("wgrad_fused" in l) or ("tile_adam" in l)
    if show: print(l)
PY
