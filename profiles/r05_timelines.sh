#!/bin/bash
# kernel tables + timelines of the office0 / scannet / indoor iterations on the current tree
cd "$(dirname "$0")/.."; export TMPDIR=/tmp; mkdir -p gpurun_out
for c in office0 scannet indoor; do
  rm -rf /tmp/pq; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pq -o t -- python bench.py --config $c --steps 120 --warmup 20 --cpu-iters 0 --no-variants > /dev/null 2>&1
  db=$(find /tmp/pq -name '*.db' | head -1)
  python profiles/summarize_rocprof_db.py $db 120 2>&1 | head -18 | cut -c1-170 > gpurun_out/r05_kernel_stats_$c.txt; cat gpurun_out/r05_kernel_stats_$c.txt
  python profiles/timeline.py $db 2>&1 | head -30 > gpurun_out/r05_timeline_$c.txt; cat gpurun_out/r05_timeline_$c.txt
done
