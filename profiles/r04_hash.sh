#!/bin/bash
# round 4: hash-grid workload (configs[1] literal) after the table-update rewrite: GPU tests, bench, kernel table, HBM counters
cd "$(dirname "$0")/.."; REPO=$PWD; export TMPDIR=/tmp; OUT=$REPO/gpurun_out/r04_hash; mkdir -p $OUT
( MNE_PARITY_STATS=$OUT/adam_stats.jsonl timeout 900 python -m pytest tests -m gpu -q -x -k "hash or grid" ) 2>&1 | tail -4
B="python $REPO/bench.py --config office0_hash --no-variants --cpu-iters 0"
timeout 300 $B --steps 300 --warmup 30 > $OUT/bench_hash.json 2> $OUT/bench_hash.err; cut -c1-1800 $OUT/bench_hash.json; tail -3 $OUT/bench_hash.err
cd /tmp
rm -rf /tmp/ks_h; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ks_h -o k -- $B --steps 100 --warmup 20 > $OUT/ks.log 2>&1
d=$(find /tmp/ks_h -name '*.db' | head -1)
python $REPO/profiles/summarize_rocprof_db.py $d > $OUT/kernel_stats_hash.txt 2>&1; head -24 $OUT/kernel_stats_hash.txt | cut -c1-170
python $REPO/profiles/timeline.py $d 12 40 > $OUT/timeline_hash.txt 2>&1
i=0
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1)); rm -rf /tmp/pmc_h$i
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc_h$i -o p -- $B --steps 10 --warmup 3 > /dev/null 2> $OUT/pmc_$i.err
done
python $REPO/profiles/pmc_traffic.py $(find /tmp/pmc_h1 -name '*.db' | head -1) $(find /tmp/pmc_h2 -name '*.db' | head -1) $OUT/pmc_traffic_hash.json $OUT/pmc_traffic_hash.txt; head -16 $OUT/pmc_traffic_hash.txt | cut -c1-150
