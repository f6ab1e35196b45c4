#!/bin/bash
# HBM traffic of the iteration's kernels: FETCH_SIZE and WRITE_SIZE in separate --pmc passes (kernel-trace only).
cd "$(dirname "$0")/.."; export TMPDIR=/tmp; mkdir -p gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_$c -o p -- python bench.py --steps 10 --warmup 3 --cpu-iters 0 > /dev/null 2> gpurun_out/pmc_$c.err
done
python profiles/pmc_traffic.py $(find /tmp/pmc_FETCH_SIZE -name '*.db' | head -1) $(find /tmp/pmc_WRITE_SIZE -name '*.db' | head -1) gpurun_out/pmc_traffic.json gpurun_out/pmc_traffic.txt
