#!/bin/bash
# hash-grid table update after a change: GPU tests of the hash / grid cases, stand-alone per-kernel timing (last 20 launches) of the shipped
# build and of any variant builds named on the command line, bench
cd "$(dirname "$0")/.."; REPO=$PWD; export TMPDIR=/tmp; OUT=$REPO/gpurun_out/r04_hash3; mkdir -p $OUT
( timeout 900 python -m pytest tests -m gpu -q -x -k "hash or grid" ) 2>&1 | tail -3
cd /tmp
for v in main "$@"; do
  lib=main; [ $v != main ] && lib=$REPO/mneslam_amd/_fuzz/$v/libmneslam_hip.so
  rm -rf /tmp/ks_a; timeout 300 rocprofv3 --kernel-trace -d /tmp/ks_a -o k -- python $REPO/profiles/r04_hash_ablate.py $lib > $OUT/ks_$v.log 2>&1
  echo "== $v: $(grep 'us per table' $OUT/ks_$v.log | cut -c17-60)"; python $REPO/profiles/last_calls.py $(find /tmp/ks_a -name '*.db' | head -1) 20 hash_ | tee $OUT/kernels_standalone_$v.txt
done
cd $REPO
bash profiles/r04_ab_variants.sh "--config office0_hash --steps 300 --warmup 30" 2 main "$@"
