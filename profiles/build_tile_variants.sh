#!/bin/bash
# Build libmneslam_hip.so variants that differ only in the tile_adam.hip tuning macros (experiments).
set -e
cd "$(dirname "$0")/.."
python -m mneslam_amd.build > /dev/null
mkdir -p profiles/_variants
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -I include -I mneslam_amd/csrc"
OBJS="mneslam_amd/csrc/capi.o mneslam_amd/csrc/render.o mneslam_amd/csrc/wgrad.o mneslam_amd/csrc/adam.o mneslam_amd/csrc/sampler.o"
for v in "1024 512 8" "512 256 8" "512 512 8" "512 256 16" "1024 256 4"; do
  set -- $v
  tag="t$1_p$2_q$3"
  hipcc $FL -DTILE_THREADS=$1 -DPASS_ENTRIES=$2 -DQB=$3 -c mneslam_amd/csrc/tile_adam.hip -o profiles/_variants/tile_$tag.o
  hipcc --offload-arch=gfx950 -shared -fPIC $OBJS profiles/_variants/tile_$tag.o -o profiles/_variants/lib_$tag.so
  echo built $tag
done
