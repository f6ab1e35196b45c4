#!/bin/bash
# round 5, first GPU pass: parity tests, A/B of the in-kernel decoder weight gradients, kernel table + timeline
cd "$(dirname "$0")/.."; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
for wg in 0 1 0 1; do
  MNE_WGRAD_INLINE=$wg python bench.py --steps 200 --warmup 30 --cpu-iters 0 --no-variants 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('inline=$wg ms/step %.4f it/s %.1f | %s %.3f | %s' % (d['ms_per_step'], d['value'], r['kernel'][:12], r['avg_launch_ms'], r['other_kernels_avg_ms']))"
done
for wg in 0 1; do
  MNE_WGRAD_INLINE=$wg python bench.py --steps 20 --warmup 5 --cpu-iters 0 --no-variants 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('driver form inline=$wg it/s %.1f' % d['value'])"
done
for wg in 0 1; do
  rm -rf /tmp/pq; MNE_WGRAD_INLINE=$wg timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/pq -o t -- python bench.py --steps 120 --warmup 20 --cpu-iters 0 --no-variants > /dev/null 2>&1
  db=$(find /tmp/pq -name '*.db' | head -1)
  python profiles/summarize_rocprof_db.py $db 120 2>&1 | head -20 | cut -c1-170 > gpurun_out/r05_first_stats_inline$wg.txt; cat gpurun_out/r05_first_stats_inline$wg.txt
  python profiles/timeline.py $db 2>&1 | head -30 > gpurun_out/r05_first_timeline_inline$wg.txt; cat gpurun_out/r05_first_timeline_inline$wg.txt
done
