#!/bin/bash
# hash scatter: per-(sample, corner) atomics vs run-reduced atomics, and where the time goes by level
cd "$(dirname "$0")/.."; export TMPDIR=/tmp; mkdir -p gpurun_out
out=gpurun_out/hash_scatter.txt; : > $out
timeout 600 python -m pytest tests/test_hip_parity_gpu.py -q -x -k "hash_grid" 2>&1 | tail -3
prof() {  # label, env...
  label=$1; shift
  rm -rf /tmp/pf; env "$@" timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/pf -o t -- python bench.py --config office0_hash --steps 40 --warmup 10 --cpu-iters 0 > /dev/null 2>&1
  db=$(find /tmp/pf -name '*.db' | head -1)
  echo "== $label" >> $out
  python profiles/summarize_rocprof_db.py $db 50 2>/dev/null | grep -E "hash_|us/iter" | cut -c1-150 >> $out
}
prof "impl 1 (atomics per sample and corner), all 16 levels" MNE_HASH_SCATTER=1
prof "impl 2 (run-reduced), all 16 levels" MNE_HASH_SCATTER=2
for n in 2 4 8 12; do prof "impl 1, first $n levels only (invalid training, timing only)" MNE_HASH_SCATTER=1 MNE_HASH_LEVELS=$n; prof "impl 2, first $n levels only" MNE_HASH_SCATTER=2 MNE_HASH_LEVELS=$n; done
cat $out
timeout 300 python bench.py --config office0_hash --steps 200 --warmup 20 --cpu-iters 0 2>/dev/null | tail -1 | cut -c1-1800
