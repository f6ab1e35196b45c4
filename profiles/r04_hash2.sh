#!/bin/bash
# round 4, second half: the hash-grid iteration with the row sort moved off the critical path (mne_hash_prebin on its own stream):
# GPU tests of the hash / grid cases, bench (twice), kernel table + timeline
cd "$(dirname "$0")/.."; REPO=$PWD; export TMPDIR=/tmp; OUT=$REPO/gpurun_out/r04_hash2; mkdir -p $OUT
( timeout 900 python -m pytest tests -m gpu -q -x -k "hash or grid" ) 2>&1 | tail -4
B="python $REPO/bench.py --config office0_hash --no-variants --cpu-iters 0"
for i in 1 2; do timeout 300 $B --steps 300 --warmup 30 > $OUT/bench_hash_$i.json 2> $OUT/bench_hash.err; cut -c1-400 $OUT/bench_hash_$i.json; done
cd /tmp
rm -rf /tmp/ks_h; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ks_h -o k -- $B --steps 100 --warmup 20 > $OUT/ks.log 2>&1
d=$(find /tmp/ks_h -name '*.db' | head -1)
python $REPO/profiles/summarize_rocprof_db.py $d > $OUT/kernel_stats_hash.txt 2>&1; head -16 $OUT/kernel_stats_hash.txt | cut -c1-170
python $REPO/profiles/timeline.py $d 12 40 > $OUT/timeline_hash.txt 2>&1; head -30 $OUT/timeline_hash.txt
