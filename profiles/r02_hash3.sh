#!/bin/bash
# hash-grid iteration with the slice update (LDS slices + fused Adam) vs run-reduced atomics
cd "$(dirname "$0")/.."; export TMPDIR=/tmp; mkdir -p gpurun_out
out=gpurun_out/hash_slices.txt; : > $out
timeout 600 python -m pytest tests/test_hip_parity_gpu.py -q -x -k "hash_grid" 2>&1 | tail -3
MNE_HASH_UPDATE=atomics timeout 600 python -m pytest tests/test_hip_parity_gpu.py -q -x -k "hash_grid" 2>&1 | tail -3
prof() {  # label, env...
  label=$1; shift
  rm -rf /tmp/pf; env "$@" timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/pf -o t -- python bench.py --config office0_hash --steps 60 --warmup 10 --cpu-iters 0 > /dev/null 2>&1
  db=$(find /tmp/pf -name '*.db' | head -1)
  echo "== $label" >> $out
  python profiles/summarize_rocprof_db.py $db 70 2>/dev/null | head -14 | cut -c1-150 >> $out
  python profiles/timeline.py $db 40 40 pack_decoder >> $out 2>&1
}
prof "slices" MNE_HASH_UPDATE=slices
prof "atomics (run-reduced)" MNE_HASH_UPDATE=atomics
cat $out
for m in slices atomics; do MNE_HASH_UPDATE=$m timeout 300 python bench.py --config office0_hash --steps 200 --warmup 20 --cpu-iters 0 2>/dev/null | tail -1 | cut -c1-330; done
