#!/bin/bash
cd "$(dirname "$0")/.."; export TMPDIR=/tmp; mkdir -p gpurun_out
out=gpurun_out/hash_slices_variants.txt; : > $out
cp mneslam_amd/libmneslam_hip.so /tmp/lib_orig.so
prof() {
  rm -rf /tmp/pf; timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/pf -o t -- python bench.py --config office0_hash --steps 30 --warmup 5 --cpu-iters 0 > /dev/null 2>&1
  db=$(find /tmp/pf -name '*.db' | head -1)
  echo "== $1" >> $out
  python profiles/summarize_rocprof_db.py $db 35 2>/dev/null | grep -E "hash_slice" | cut -c1-150 >> $out
}
prof default
for lib in profiles/_variants/lib_*.so; do cp $lib mneslam_amd/libmneslam_hip.so; prof $lib; done
cp /tmp/lib_orig.so mneslam_amd/libmneslam_hip.so
cat $out
