#!/bin/bash
# Build libmneslam_hip.so variants that differ only in render.hip tuning macros (experiments).
# usage: bash profiles/build_render_variants.sh "tag1:-DA=1 -DB=2" "tag2:-DA=3" ...
set -e
cd "$(dirname "$0")/.."
python -m mneslam_amd.build > /dev/null
rm -rf profiles/_variants; mkdir -p profiles/_variants
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -I include -I mneslam_amd/csrc"
OBJS="mneslam_amd/csrc/capi.o mneslam_amd/csrc/wgrad.o mneslam_amd/csrc/adam.o mneslam_amd/csrc/sampler.o mneslam_amd/csrc/tile_adam.o mneslam_amd/csrc/gridenc.o"
for v in "$@"; do
  tag="${v%%:*}"; defs="${v#*:}"
  hipcc $FL $defs -c mneslam_amd/csrc/render.hip -o profiles/_variants/render_$tag.o
  hipcc --offload-arch=gfx950 -shared -fPIC $OBJS profiles/_variants/render_$tag.o -o profiles/_variants/lib_$tag.so
  echo built $tag
done
