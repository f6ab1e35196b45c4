#!/bin/bash
set -e
cd "$(dirname "$0")/.."
python -m mneslam_amd.build > /dev/null
mkdir -p profiles/_variants
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -I include -I mneslam_amd/csrc"
OBJS="mneslam_amd/csrc/capi.o mneslam_amd/csrc/render.o mneslam_amd/csrc/wgrad.o mneslam_amd/csrc/adam.o mneslam_amd/csrc/sampler.o mneslam_amd/csrc/gridenc.o"
hipcc $FL -DTILE_PROFILE "$@" -c mneslam_amd/csrc/tile_adam.hip -o profiles/_variants/tile_profile.o
hipcc --offload-arch=gfx950 -shared -fPIC $OBJS profiles/_variants/tile_profile.o -o profiles/_variants/libprofile.so
