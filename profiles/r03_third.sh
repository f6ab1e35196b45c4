#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r03_third
mkdir -p $OUT
export PYTHONPATH=$PWD
echo "== bench, driver form" | tee $OUT/bench.txt
python bench.py --steps 20 --warmup 5 --cpu-iters 0 2>&1 | tail -1 | tee -a $OUT/bench.txt
echo "== bench, driver form, MNE_NO_ADAPT=1" | tee -a $OUT/bench.txt
MNE_NO_ADAPT=1 python bench.py --steps 20 --warmup 5 --cpu-iters 0 2>&1 | tail -1 | tee -a $OUT/bench.txt
echo "== bench, 200 steps" | tee -a $OUT/bench.txt
python bench.py --steps 200 --warmup 20 --cpu-iters 0 2>&1 | tail -1 | tee -a $OUT/bench.txt
timeout 1700 python -m pytest tests -m gpu -q -x --durations=15 2>&1 | tail -60 > $OUT/pytest.txt
tail -5 $OUT/pytest.txt
