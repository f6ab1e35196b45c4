#!/bin/bash
# the two bench lines of the final tree (after the event cadence change): default (variants + CPU baseline) and 3x driver form
cd "$(dirname "$0")/.."; export TMPDIR=/tmp; mkdir -p gpurun_out
rm -f gpurun_out/r03_bench_driver_form.json
for i in 1 2 3; do timeout 300 python bench.py --steps 20 --warmup 5 --cpu-iters 0 --no-variants >> gpurun_out/r03_bench_driver_form.json 2>/dev/null; done; cut -c1-200 gpurun_out/r03_bench_driver_form.json
timeout 600 python bench.py > gpurun_out/r03_bench_default.json 2> gpurun_out/r03_bench_default.err; cut -c1-260 gpurun_out/r03_bench_default.json
