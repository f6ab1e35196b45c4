#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r03_arrange; mkdir -p $OUT; export PYTHONPATH=$PWD
run() { python bench.py "$@" --cpu-iters 0 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('  %.1f it/s %.4f ms | %s %.3f |'%(d['value'],d['ms_per_step'],r['kernel'][:10],r['avg_launch_ms']), {k[:18]:round(v,3) for k,v in r['other_kernels_avg_ms'].items()})"; }
for a in split critical split critical; do echo "== 200 steps MNE_ARRANGE=$a" | tee -a $OUT/out.txt; MNE_ARRANGE=$a run --steps 200 --warmup 20 | tee -a $OUT/out.txt; done
for a in split critical; do echo "== driver form MNE_ARRANGE=$a" | tee -a $OUT/out.txt; MNE_ARRANGE=$a run --steps 20 --warmup 5 | tee -a $OUT/out.txt; done
echo "== no event timing, 200 steps" | tee -a $OUT/out.txt
for a in split critical; do MNE_ARRANGE=$a run --steps 200 --warmup 20 --event-every 100000 | tee -a $OUT/out.txt; done
MNE_ARRANGE=critical timeout 900 python -m pytest tests/test_hip_parity_gpu.py -m gpu -q -k "full_size or baseline_config or three_fused or rebinding or batch_shapes or quality" 2>&1 | tail -4 | tee -a $OUT/out.txt
