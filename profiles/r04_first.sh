#!/bin/bash
# round 4, first GPU pass: GPU parity suite (+ measured Adam agreement stats), smoke, driver-form bench with variants
cd "$(dirname "$0")/.."; export TMPDIR=/tmp; mkdir -p gpurun_out; OUT=gpurun_out/r04_first; mkdir -p $OUT
rm -f $OUT/adam_stats.jsonl
( time MNE_PARITY_STATS=$PWD/$OUT/adam_stats.jsonl timeout 1500 python -m pytest tests -m gpu -q -x --durations=15 ) > $OUT/pytest.txt 2>&1; tail -25 $OUT/pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_form.json 2> $OUT/bench_driver_form.err; tail -c 6000 $OUT/bench_driver_form.json; tail -5 $OUT/bench_driver_form.err
