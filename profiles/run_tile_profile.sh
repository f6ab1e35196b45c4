#!/bin/bash
cd "$(dirname "$0")/.."; export MNE_NO_OVERLAP=1
cp mneslam_amd/libmneslam_hip.so /tmp/lib_orig.so
for lib in profiles/_variants/libprofile_*.so; do
  cp $lib mneslam_amd/libmneslam_hip.so
  echo "== $lib"
  python profiles/tile_phase_times.py 2>&1 | grep -E "blocks|entries|percentiles" | cut -c1-260
done
cp /tmp/lib_orig.so mneslam_amd/libmneslam_hip.so
