#!/bin/bash
cd "$(dirname "$0")/.."; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x -k "graph or clock or three_fused or full_size_fused" 2>&1 | tail -3
for g in 1 0; do
  MNE_GRAPH=$g python bench.py --steps 300 --warmup 30 --cpu-iters 0 2>gpurun_out/bench_graph$g.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('GRAPH=$g ms/step %.4f it/s %.1f psnr %.2f | %s %.3f | %s' % (d['ms_per_step'], d['value'], d['psnr_last_iter'], r['kernel'][:12], r['avg_launch_ms'], list(r['other_kernels_avg_ms'].values())))"
  tail -3 gpurun_out/bench_graph$g.err
done
MNE_GRAPH=1 python bench.py --steps 300 --warmup 30 --cpu-iters 0 --event-every 100000 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('graph, no events: ms/step %.4f it/s %.1f psnr %.2f' % (d['ms_per_step'], d['value'], d['psnr_last_iter']))"
rm -rf /tmp/pv; MNE_GRAPH=1 timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/pv -o t -- python bench.py --steps 60 --warmup 10 --cpu-iters 0 --event-every 100000 > /dev/null 2>&1
db=$(find /tmp/pv -name '*.db' | head -1)
python profiles/summarize_rocprof_db.py $db 70 2>&1 | head -18 | cut -c1-150
python profiles/timeline.py $db 30 20 gather_kernel
