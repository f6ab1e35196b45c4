#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r03_fifth
mkdir -p $OUT
export PYTHONPATH=$PWD
run() { python bench.py "$@" --cpu-iters 0 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('  %.1f it/s %.4f ms | tile_adam %.3f |'%(d['value'],d['ms_per_step'],r['avg_launch_ms']), {k[:22]:round(v,3) for k,v in r['other_kernels_avg_ms'].items()})"; }
echo "== 200 steps" | tee -a $OUT/bench.txt; run --steps 200 --warmup 20 | tee -a $OUT/bench.txt
echo "== 200 steps, empty lists (pure Adam sweep)" | tee -a $OUT/bench.txt; MNE_ABL_EMPTY_LISTS=1 run --steps 200 --warmup 20 | tee -a $OUT/bench.txt
echo "== 200 steps, --no-overlap (one stream: kernels timed alone)" | tee -a $OUT/bench.txt; run --steps 200 --warmup 20 --no-overlap | tee -a $OUT/bench.txt
echo "== 200 steps, --no-overlap, empty lists" | tee -a $OUT/bench.txt; MNE_ABL_EMPTY_LISTS=1 run --steps 200 --warmup 20 --no-overlap | tee -a $OUT/bench.txt
for c in scannet indoor apartment office0_hash; do echo "== $c" | tee -a $OUT/bench.txt; run --steps 100 --warmup 20 --config $c | tee -a $OUT/bench.txt; done
echo "== office0 --hidden 64" | tee -a $OUT/bench.txt; run --steps 100 --warmup 20 --hidden 64 | tee -a $OUT/bench.txt
echo "== scannet --hidden 64" | tee -a $OUT/bench.txt; run --steps 100 --warmup 20 --config scannet --hidden 64 | tee -a $OUT/bench.txt
