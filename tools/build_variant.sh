#!/bin/bash
# Development A/B: build a variant of some translation units with extra -D flags and link it with the other (already built)
# objects into scaleprotoseg_amd/variants/libspx_<name>.so (selected at run time with SPX_LIB_OVERRIDE; the product library is
# never touched).   tools/build_variant.sh <name> "<tu> [<tu> ...]" [flags...]      e.g.  ... nobar "spx_bwd_npb6" -DSPX_DIAG_X
set -e
cd "$(dirname "$0")/../scaleprotoseg_amd"
name=$1; tus=$2; shift 2
mkdir -p variants csrc/build/var_$name
skip=""
pids=""
for tu in $tus; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off "$@" -c csrc/$tu.hip -o csrc/build/var_$name/$tu.o &
  pids="$pids $!"
  skip="$skip -e /$tu.o"
done
for p in $pids; do wait $p; done
objs=$(ls csrc/build/*.o | grep -v $skip)
hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libspx_$name.so csrc/build/var_$name/*.o $objs
echo variants/libspx_$name.so
