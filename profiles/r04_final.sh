#!/bin/bash
# round 4, final GPU pass: the whole GPU suite, smoke, the default bench line and three driver-form lines, then the hash-grid workload's
# kernel table / timeline / HBM counter passes on the final kernels (the tri-plane kernels' passes: profiles/r04_pmc.sh, unchanged since)
cd "$(dirname "$0")/.."; REPO=$PWD; export TMPDIR=/tmp; OUT=$REPO/gpurun_out/r04_final; mkdir -p $OUT
( time timeout 1500 python -m pytest tests -m gpu -q -x --durations=8 ) > $OUT/pytest.txt 2>&1; tail -14 $OUT/pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; cut -c1-700 $OUT/bench_default.json; tail -2 $OUT/bench_default.err
for i in 1 2 3; do timeout 600 python bench.py --steps 20 --warmup 5 $([ $i != 1 ] && echo --no-variants --cpu-iters 0) > $OUT/bench_driver_form_$i.json 2>> $OUT/bench_driver.err; cut -c1-160 $OUT/bench_driver_form_$i.json; done
B="python $REPO/bench.py --config office0_hash --no-variants --cpu-iters 0"
cd /tmp
# configs[4] as a replayed graph: the kernel table of that variant (events cannot be recorded inside a graph)
rm -rf /tmp/ks_g; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ks_g -o k -- python $REPO/bench.py --config indoor_fp16 --graph one_stream --no-variants --cpu-iters 0 --steps 100 --warmup 20 > $OUT/ks_indoor_fp16_graph.log 2>&1
python $REPO/profiles/summarize_rocprof_db.py $(find /tmp/ks_g -name '*.db' | head -1) > $OUT/kernel_stats_indoor_fp16_graph.txt 2>&1; head -8 $OUT/kernel_stats_indoor_fp16_graph.txt | cut -c1-150
rm -rf /tmp/ks_h; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ks_h -o k -- $B --steps 100 --warmup 20 > $OUT/ks_hash.log 2>&1
d=$(find /tmp/ks_h -name '*.db' | head -1)
python $REPO/profiles/summarize_rocprof_db.py $d > $OUT/hash_kernel_stats.txt 2>&1; head -14 $OUT/hash_kernel_stats.txt | cut -c1-170
python $REPO/profiles/timeline.py $d 12 40 > $OUT/hash_timeline.txt 2>&1
i=0
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1)); rm -rf /tmp/pmc_h$i
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc_h$i -o p -- $B --steps 10 --warmup 3 > /dev/null 2> $OUT/pmc_$i.err
done
python $REPO/profiles/pmc_traffic.py $(find /tmp/pmc_h1 -name '*.db' | head -1) $(find /tmp/pmc_h2 -name '*.db' | head -1) $OUT/hash_pmc_traffic.json $OUT/hash_pmc_traffic.txt; head -12 $OUT/hash_pmc_traffic.txt | cut -c1-150
