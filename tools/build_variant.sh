#!/bin/bash
# Development A/B: build a variant of the fused-backward translation unit with extra -D flags and link it with the other
# (already built) objects into scaleprotoseg_amd/variants/libspx_<name>.so.   tools/build_variant.sh <name> [flags...]
set -e
cd "$(dirname "$0")/../scaleprotoseg_amd"
name=$1; shift
mkdir -p variants csrc/build/var_$name
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off "$@" -c csrc/spx_bwdf.hip -o csrc/build/var_$name/spx_bwdf.o
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off "$@" -c csrc/spx_api.hip -o csrc/build/var_$name/spx_api.o
objs=$(ls csrc/build/*.o | grep -v "spx_bwdf.o\|spx_api.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libspx_$name.so csrc/build/var_$name/spx_bwdf.o csrc/build/var_$name/spx_api.o $objs
echo variants/libspx_$name.so
