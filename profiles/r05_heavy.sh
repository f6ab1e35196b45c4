#!/bin/bash
# heavy-ray list + tile-parallel backward (heavy_bwd_kernel): A/B on INS Indoor (the config with long rays), parity test, timeline
cd "$(dirname "$0")/.."; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_parity_gpu.py -m gpu -q -k "indoor or configs4 or apartment" 2>&1 | tail -4
for c in indoor indoor_fp16; do for v in noheavy main noheavy main; do
  python profiles/r03_variant_bench.py $v --config $c --steps 200 --warmup 30 --cpu-iters 0 --no-variants 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
o = dict(r['other_kernels_avg_ms']); o[r['kernel']] = r['avg_launch_ms']
print('$c $v ms/step %.4f it/s %.1f | ' % (d['ms_per_step'], d['value']) + ' '.join('%s=%.3f' % (k.split(' ')[0], v) for k, v in o.items()))"
done; done
rm -rf /tmp/pq; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pq -o t -- python bench.py --config indoor --steps 120 --warmup 20 --cpu-iters 0 --no-variants > /dev/null 2>&1
db=$(find /tmp/pq -name '*.db' | head -1)
python profiles/summarize_rocprof_db.py $db 120 2>&1 | head -16 | cut -c1-170 > gpurun_out/r05_kernel_stats_indoor_heavy.txt; cat gpurun_out/r05_kernel_stats_indoor_heavy.txt
python profiles/timeline.py $db 2>&1 | head -30 > gpurun_out/r05_timeline_indoor_heavy.txt; cat gpurun_out/r05_timeline_indoor_heavy.txt
for v in noheavy main; do
  python profiles/r03_variant_bench.py $v --steps 200 --warmup 30 --cpu-iters 0 --no-variants 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('office0 $v it/s %.1f' % d['value'])"
done
