#!/bin/bash
# Build libmneslam_hip.so variants that differ only in the tuning macros of ONE source file (experiments).
# usage: bash profiles/build_variants.sh <file.hip> "tag1:-DA=1 -DB=2" "tag2:-DA=3" ...
set -e
cd "$(dirname "$0")/.."
src=$1; shift
python -m mneslam_amd.build > /dev/null
rm -rf profiles/_variants; mkdir -p profiles/_variants
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -I include -I mneslam_amd/csrc"
OBJS=""
for f in capi render wgrad adam sampler tile_adam gridenc; do [ "$f.hip" != "$src" ] && OBJS="$OBJS mneslam_amd/csrc/$f.o"; done
for v in "$@"; do
  tag="${v%%:*}"; defs="${v#*:}"
  hipcc $FL $defs -c mneslam_amd/csrc/$src -o profiles/_variants/v_$tag.o
  hipcc --offload-arch=gfx950 -shared -fPIC $OBJS profiles/_variants/v_$tag.o -o profiles/_variants/lib_$tag.so
  echo built $tag
done
