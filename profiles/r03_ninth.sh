#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r03_ninth; mkdir -p $OUT; export PYTHONPATH=$PWD
run() { python bench.py "$@" --cpu-iters 0 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('  %.1f it/s %.4f ms | %s %.3f |'%(d['value'],d['ms_per_step'],r['kernel'][:10],r['avg_launch_ms']), {k[:18]:round(v,3) for k,v in r['other_kernels_avg_ms'].items()})"; }
for e in 0 1 0 1; do echo "== 200 steps MNE_TORCH_EVENTS=$e" | tee -a $OUT/out.txt; MNE_TORCH_EVENTS=$e run --steps 200 --warmup 20 | tee -a $OUT/out.txt; done
for e in 0 1; do echo "== driver form MNE_TORCH_EVENTS=$e" | tee -a $OUT/out.txt; MNE_TORCH_EVENTS=$e run --steps 20 --warmup 5 | tee -a $OUT/out.txt; done
timeout 900 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|Error" | tail -4 | tee -a $OUT/out.txt
