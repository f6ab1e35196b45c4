#!/bin/bash
# Round 3, GPU call 1: (a) the driver-form and the 200-step bench line of the library as round 2 left it, (b) the
# layout-fuzz root-cause matrix for ray_kernel<64,64,colour planes,4>, (c) the LDS-atomic microbenchmark.
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r03_first
mkdir -p $OUT
export PYTHONPATH=$PWD
echo "== bench, driver form (--steps 20 --warmup 5)" | tee $OUT/bench.txt
python bench.py --steps 20 --warmup 5 --cpu-iters 0 2>&1 | tail -1 | tee -a $OUT/bench.txt
echo "== bench, 200 steps" | tee -a $OUT/bench.txt
python bench.py --steps 200 --warmup 20 --cpu-iters 0 2>&1 | tail -1 | tee -a $OUT/bench.txt
echo "== lds atomic microbenchmark" | tee $OUT/lds_atomic.txt
./profiles/_bin/lds_atomic_bench 2>&1 | tee -a $OUT/lds_atomic.txt
: > $OUT/fuzz.txt
for v in main pad16 pad8 x_pad16_syncvm x_pad16_wpb8 x_pad16_wpb4 x_pad16_noseq; do
  lib=mneslam_amd/_fuzz/$v/libmneslam_hip.so
  [ $v = main ] && lib=mneslam_amd/libmneslam_hip.so
  echo "==== $v" | tee -a $OUT/fuzz.txt
  timeout 300 python profiles/r03_layout_fuzz_diag.py $lib 2>&1 | grep -v "^$" | tail -24 | tee -a $OUT/fuzz.txt
done
for v in main pad16; do
  lib=mneslam_amd/_fuzz/$v/libmneslam_hip.so
  [ $v = main ] && lib=mneslam_amd/libmneslam_hip.so
  echo "==== $v --poison" | tee -a $OUT/fuzz.txt
  timeout 300 python profiles/r03_layout_fuzz_diag.py $lib --poison 2>&1 | grep -v "^$" | tail -24 | tee -a $OUT/fuzz.txt
done
echo done
