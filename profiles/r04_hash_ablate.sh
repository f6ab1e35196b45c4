#!/bin/bash
# timing-only ablation builds of the hash-grid table update (mneslam_amd/_fuzz/abl_*: -DHASH_ABL=bits of profiles/patches/
# r04_hash_ablation_switch.patch), stand-alone, per kernel: the last 20 launches of each run are the stand-alone ones
cd "$(dirname "$0")/.."; REPO=$PWD; export TMPDIR=/tmp; OUT=$REPO/gpurun_out/r04_hash_ablate; mkdir -p $OUT
F=$REPO/mneslam_amd/_fuzz
cd /tmp
for v in main "$@"; do
  lib=main; [ $v != main ] && lib=$F/$v/libmneslam_hip.so
  rm -rf /tmp/ks_a; timeout 300 rocprofv3 --kernel-trace -d /tmp/ks_a -o k -- python $REPO/profiles/r04_hash_ablate.py $lib > $OUT/ks_$v.log 2>&1
  echo "== $v: $(grep 'us per table' $OUT/ks_$v.log | cut -c1-60)"
  python $REPO/profiles/last_calls.py $(find /tmp/ks_a -name '*.db' | head -1) 20 hash_
done
