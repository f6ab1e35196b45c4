#!/bin/bash
cd "$(dirname "$0")/.."; export TMPDIR=/tmp; mkdir -p gpurun_out
out=gpurun_out/hash_masks.txt; : > $out
timeout 600 python -m pytest tests/test_hip_parity_gpu.py -q -x -k "hash_grid" 2>&1 | grep -E "passed|failed" >> $out
prof() {
  label=$1; shift
  rm -rf /tmp/pf; env "$@" timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/pf -o t -- python bench.py --config office0_hash --steps 30 --warmup 5 --cpu-iters 0 > /dev/null 2>&1
  db=$(find /tmp/pf -name '*.db' | head -1)
  echo "== $label" >> $out
  python profiles/summarize_rocprof_db.py $db 35 2>/dev/null | grep -E "hash_slice|hash_pack" | cut -c1-150 >> $out
}
prof "masks, 16 levels" A=1
prof "masks, first 7 levels (dense only)" MNE_HASH_LEVELS=7
prof "masks, first 10 levels" MNE_HASH_LEVELS=10
prof "no masks, 16 levels" MNE_HASH_NO_MASKS=1
MNE_NO_OVERLAP=1 timeout 300 python bench.py --config office0_hash --steps 200 --warmup 20 --cpu-iters 0 2>/dev/null | tail -1 | cut -c1-200 >> $out
timeout 300 python bench.py --config office0_hash --steps 200 --warmup 20 --cpu-iters 0 2>/dev/null | tail -1 | cut -c1-200 >> $out
cat $out
