#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r03_fourth
mkdir -p $OUT
export PYTHONPATH=$PWD
run() { python bench.py "$@" --cpu-iters 0 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('  %.1f it/s %.4f ms | tile_adam %.3f |'%(d['value'],d['ms_per_step'],r['avg_launch_ms']), {k[:22]:round(v,3) for k,v in r['other_kernels_avg_ms'].items()})"; }
echo "== driver form" | tee -a $OUT/bench.txt; run --steps 20 --warmup 5 | tee -a $OUT/bench.txt
echo "== driver form, serial bin" | tee -a $OUT/bench.txt; MNE_SERIAL_BIN=1 run --steps 20 --warmup 5 | tee -a $OUT/bench.txt
echo "== 200 steps" | tee -a $OUT/bench.txt; run --steps 200 --warmup 20 | tee -a $OUT/bench.txt
echo "== 200 steps, serial bin" | tee -a $OUT/bench.txt; MNE_SERIAL_BIN=1 run --steps 200 --warmup 20 | tee -a $OUT/bench.txt
echo "== 200 steps again" | tee -a $OUT/bench.txt; run --steps 200 --warmup 20 | tee -a $OUT/bench.txt
timeout 1700 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > $OUT/pytest.txt
tail -3 $OUT/pytest.txt
