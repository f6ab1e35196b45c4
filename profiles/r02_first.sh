#!/bin/bash
# round 2, first GPU pass: parity tests (new: corner indices, full-size oracle parity, C3/C4/C5 shapes), bench, kernel table,
# tile_adam prefetch variants
cd "$(dirname "$0")/.."; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x --durations=15 > gpurun_out/pytest_gpu.log 2>&1; tail -30 gpurun_out/pytest_gpu.log
timeout 600 python bench.py --cpu-iters 0 > gpurun_out/bench_r02_a.json 2> gpurun_out/bench_r02_a.err; tail -c 1500 gpurun_out/bench_r02_a.json
for c in apartment scannet indoor; do
  timeout 600 python bench.py --config $c --steps 50 --warmup 10 --cpu-iters 0 2> gpurun_out/bench_$c.err | tail -1 | cut -c1-700
done
bash profiles/run_variants.sh
