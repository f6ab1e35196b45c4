#!/bin/bash
cd "$(dirname "$0")/.."; export TMPDIR=/tmp; mkdir -p gpurun_out
out=gpurun_out/hash_slices_levels.txt; : > $out
timeout 600 python -m pytest tests/test_hip_parity_gpu.py -q -x -k "hash_grid" 2>&1 | tail -3
prof() {  # label, env...
  label=$1; shift
  rm -rf /tmp/pf; env "$@" timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/pf -o t -- python bench.py --config office0_hash --steps 30 --warmup 5 --cpu-iters 0 > /dev/null 2>&1
  db=$(find /tmp/pf -name '*.db' | head -1)
  echo "== $label" >> $out
  python profiles/summarize_rocprof_db.py $db 35 2>/dev/null | grep -E "hash_slice|hash_pack|hash_dense" | cut -c1-150 >> $out
}
for n in 16 1 4 7 8 12; do prof "slices, first $n levels" MNE_HASH_LEVELS=$n; done
cat $out
timeout 300 python bench.py --config office0_hash --steps 200 --warmup 20 --cpu-iters 0 2>/dev/null | tail -1 | cut -c1-330
