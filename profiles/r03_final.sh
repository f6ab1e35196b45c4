#!/bin/bash
# round-3 measurement pass on the final tree: parity tests, smoke, default bench line (variants + CPU baseline on the
# device-drawn batch), driver-form line, counter passes + kernel tables + timelines, first-steps table of a fresh model
cd "$(dirname "$0")/.."; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r03_gpu_tests.txt 2>&1; grep -E "passed|failed|error" gpurun_out/r03_gpu_tests.txt | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python bench.py > gpurun_out/r03_bench_default.json 2> gpurun_out/r03_bench_default.err; cut -c1-260 gpurun_out/r03_bench_default.json
for i in 1 2 3; do timeout 300 python bench.py --steps 20 --warmup 5 --cpu-iters 0 --no-variants >> gpurun_out/r03_bench_driver_form.json 2>/dev/null; done; cut -c1-200 gpurun_out/r03_bench_driver_form.json
bash profiles/r03_pmc.sh
cut -c1-170 gpurun_out/r03_pmc_traffic.txt | head -16
timeout 300 python profiles/r03_step_kernels.py 30 > gpurun_out/r03_first_steps.txt 2>&1
timeout 300 python profiles/r03_step_times.py 80 >> gpurun_out/r03_first_steps.txt 2>&1; tail -3 gpurun_out/r03_first_steps.txt | cut -c1-300
