#!/bin/bash
# per-level cost of the hash-grid table update (stand-alone, HIP events), then the workload's bench / kernel table / counters
cd "$(dirname "$0")/.."; OUT=gpurun_out/r04_hash_levels; mkdir -p $OUT
timeout 200 python profiles/r04_hash_levels.py 2>&1 | grep -v amdgpu.ids | tee $OUT/levels.txt
bash profiles/r04_hash.sh
