#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r03_second
mkdir -p $OUT
export PYTHONPATH=$PWD
: > $OUT/fuzz2.txt
for v in main pad16 pad8; do
  lib=mneslam_amd/_fuzz/$v/libmneslam_hip.so
  [ $v = main ] && lib=mneslam_amd/libmneslam_hip.so
  echo "==== $v" | tee -a $OUT/fuzz2.txt
  timeout 300 python profiles/r03_layout_fuzz_diag2.py $lib 2>&1 | grep -v "amdgpu.ids" | tail -40 | tee -a $OUT/fuzz2.txt
done
echo "== driver form, no early termination (decode everything)" | tee $OUT/bench.txt
MNE_NO_EARLY_TERMINATION=1 python bench.py --steps 20 --warmup 5 --cpu-iters 0 2>&1 | tail -1 | tee -a $OUT/bench.txt
echo done
