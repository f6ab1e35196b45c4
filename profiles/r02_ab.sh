#!/bin/bash
cd "$(dirname "$0")/.."; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q -x -k "full_batch_properties" 2>&1 | tail -3
run() {  # tag, env...
  tag=$1; shift
  env "$@" python bench.py --steps 200 --warmup 30 --cpu-iters 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('$tag ms/step %.4f it/s %.1f | %s %.3f | %s' % (d['ms_per_step'], d['value'], r['kernel'][:12], r['avg_launch_ms'], list(r['other_kernels_avg_ms'].values())))"
  rm -rf /tmp/p_$tag; env "$@" timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/p_$tag -o t -- python bench.py --steps 60 --warmup 10 --cpu-iters 0 > /dev/null 2>&1
  db=$(find /tmp/p_$tag -name '*.db' | head -1)
  python profiles/summarize_rocprof_db.py $db 70 2>&1 | head -18 | cut -c1-150 > gpurun_out/stats_$tag.txt
  python profiles/timeline.py $db 30 20 > gpurun_out/timeline_$tag.txt 2>&1
  python profiles/gap_analysis.py $db > gpurun_out/gaps_$tag.txt 2>&1
  head -8 gpurun_out/stats_$tag.txt
}
run after X=0
cat gpurun_out/timeline_after.txt
run first MNE_TILE_ORDER_FIRST=1
cat gpurun_out/timeline_first.txt
cp mneslam_amd/libmneslam_hip.so /tmp/lib_orig.so
cp profiles/_variants/lib_abl.so mneslam_amd/libmneslam_hip.so
run abl_first MNE_TILE_ORDER_FIRST=1
cp /tmp/lib_orig.so mneslam_amd/libmneslam_hip.so
