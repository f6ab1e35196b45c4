#!/bin/bash
# round-2 measurement pass: parity tests, smoke, default bench, other workloads, render_img, kernel table + timeline,
# PMC traffic / SQ counters, matched-quality trajectory
cd "$(dirname "$0")/.."; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -c 3000 gpurun_out/bench_default.json
for c in "--hidden 64" "--config apartment" "--config scannet" "--config scannet --hidden 64" "--config indoor" "--scatter atomics" "--path autograd --steps 50"; do
  echo "== bench.py $c"; timeout 400 python bench.py $c --steps 100 --warmup 20 --cpu-iters 0 2>/dev/null | tail -1 | cut -c1-420
done
echo "== render_img"; timeout 400 python bench.py --mode render_img --steps 60 --warmup 20 2>gpurun_out/render_img.err | tail -1; tail -2 gpurun_out/render_img.err
rm -rf /tmp/pf; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pf -o t -- python bench.py --steps 100 --warmup 20 --cpu-iters 0 > /dev/null 2>&1
db=$(find /tmp/pf -name '*.db' | head -1)
python profiles/summarize_rocprof_db.py $db 120 > gpurun_out/kernel_stats_r02.txt 2>&1; head -22 gpurun_out/kernel_stats_r02.txt | cut -c1-160
python profiles/timeline.py $db 40 40 > gpurun_out/timeline_r02.txt 2>&1; cat gpurun_out/timeline_r02.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_$c -o p -- python bench.py --steps 10 --warmup 3 --cpu-iters 0 > /dev/null 2> gpurun_out/pmc_$c.err
done
rm -rf /tmp/pmc_sq; timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES -d /tmp/pmc_sq -o p -- python bench.py --steps 10 --warmup 3 --cpu-iters 0 > /dev/null 2> gpurun_out/pmc_sq.err
python profiles/pmc_summary.py $(find /tmp/pmc_FETCH_SIZE -name '*.db' | head -1) $(find /tmp/pmc_WRITE_SIZE -name '*.db' | head -1) $(find /tmp/pmc_sq -name '*.db' | head -1) gpurun_out/pmc_r02.json gpurun_out/pmc_r02.txt
python profiles/summarize_pmc_db.py "_kernel" $(find /tmp/pmc_sq -name '*.db' | head -1) > gpurun_out/pmc_sq_r02.txt 2>&1
timeout 900 python profiles/quality_trajectory.py 300 > gpurun_out/quality_trajectory.txt 2>&1; tail -45 gpurun_out/quality_trajectory.txt
