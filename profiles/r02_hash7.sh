#!/bin/bash
cd "$(dirname "$0")/.."; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_hip_parity_gpu.py -q -x -k "hash_grid" 2>&1 | tail -3
MNE_NO_EARLY_TERMINATION=1 timeout 600 python -m pytest tests/test_hip_parity_gpu.py -q -x -k "hash_grid and not learns" 2>&1 | tail -2
timeout 600 python bench.py --config office0_hash --steps 200 --warmup 20 > gpurun_out/bench_hash.json 2>/dev/null; tail -c 3000 gpurun_out/bench_hash.json
rm -rf /tmp/pf; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pf -o t -- python bench.py --config office0_hash --steps 100 --warmup 20 --cpu-iters 0 > /dev/null 2>&1
db=$(find /tmp/pf -name '*.db' | head -1)
python profiles/summarize_rocprof_db.py $db 120 > gpurun_out/kernel_stats_hash.txt 2>&1; head -22 gpurun_out/kernel_stats_hash.txt | cut -c1-170
python profiles/timeline.py $db 40 40 pack_decoder > gpurun_out/timeline_hash.txt 2>&1; cat gpurun_out/timeline_hash.txt
