#!/bin/bash
# hash-grid iteration (configs[1] literal): slice kernel without the chunk-group loop (un-regressed), wave priority of wgrad_fused64_kernel
cd "$(dirname "$0")/.."; REPO=$PWD; export TMPDIR=/tmp; OUT=$REPO/gpurun_out/r06_hash; mkdir -p $OUT
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value'],1), 'it/s', round(d['ms_per_step'],4), 'ms; dominant', r['kernel'][:40], round(r['avg_launch_ms']*1000,1), 'us;', {k[:22]: round(v*1000,1) for k,v in r.get('other_kernels_avg_ms',{}).items()})"; }
for rep in 1 2; do
for prio in 0 1 2 3; do
  echo -n "office0_hash 300 steps MNE_WGRAD_PRIO=$prio: " | tee -a $OUT/lines.txt
  MNE_WGRAD_PRIO=$prio timeout 300 python bench.py --config office0_hash --no-variants --cpu-iters 0 --steps 300 --warmup 50 2>/dev/null | tail -1 | line | tee -a $OUT/lines.txt
done; done
cd /tmp; rm -rf /tmp/ks_h
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ks_h -o k -- python $REPO/bench.py --config office0_hash --no-variants --cpu-iters 0 --steps 100 --warmup 20 > /tmp/ks_h.log 2>&1
db=$(find /tmp/ks_h -name '*.db' | head -1)
python $REPO/profiles/summarize_rocprof_db.py $db > $OUT/kernel_stats.txt 2>&1; head -14 $OUT/kernel_stats.txt | cut -c1-150
python $REPO/profiles/timeline.py $db 12 40 > $OUT/timeline.txt 2>&1; head -24 $OUT/timeline.txt
