#!/bin/bash
# quick GPU check: parity tests, bench with/without stream overlap, kernel table
cd "$(dirname "$0")/.."; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for ov in 0 1; do
  MNE_NO_OVERLAP=$ov python bench.py --steps 200 --warmup 30 --cpu-iters 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('no_overlap=$ov ms/step %.4f it/s %.1f | %s %.3f | %s' % (d['ms_per_step'], d['value'], r['kernel'][:12], r['avg_launch_ms'], r['other_kernels_avg_ms']))"
done
rm -rf /tmp/pq; MNE_NO_OVERLAP=${NOOV:-0} timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/pq -o t -- python bench.py --steps 60 --warmup 10 --cpu-iters 0 > /dev/null 2>&1
python profiles/summarize_rocprof_db.py $(find /tmp/pq -name '*.db' | head -1) 2>&1 | head -22 | cut -c1-150 > gpurun_out/quick_stats.txt; cat gpurun_out/quick_stats.txt
