#!/bin/bash
# kernel table + one-iteration timeline of the default workload (rocprofv3 --kernel-trace --stats)
cd "$GRAFT_REPO_ROOT" || exit 1; export TMPDIR=/tmp; export PYTHONPATH=$PWD; mkdir -p gpurun_out/r03_tl
rm -rf /tmp/pv; cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pv -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 60 --warmup 10 --cpu-iters 0 --event-every 1000 > /tmp/pv_bench.txt 2>&1
cd $GRAFT_REPO_ROOT
db=$(find /tmp/pv -name '*.db' | head -1)
python profiles/summarize_rocprof_db.py $db 70 2>&1 | head -30 | cut -c1-160 > gpurun_out/r03_tl/stats.txt
python profiles/timeline.py $db 30 20 > gpurun_out/r03_tl/timeline.txt 2>&1
tail -1 /tmp/pv_bench.txt | cut -c1-200 > gpurun_out/r03_tl/bench_line.txt
cat gpurun_out/r03_tl/stats.txt gpurun_out/r03_tl/timeline.txt
