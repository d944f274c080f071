#!/bin/bash
# A variant of the library with another row-stream filter (compile-time switches of kernels_ls.hip: -DLS_DEPTH=3, -DLS_ABL=1), for benchmarks/ab_libs.sh-style comparisons:
#   bash benchmarks/build_ls_variant.sh <name> -DLS_DEPTH=3   -> benchmarks/bin/libcanopy_<name>.so
set -e
NAME=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/benchmarks/bin
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -fvisibility=hidden -DCANOPY_BUILD -DCANOPY_SHARED \
  -I$R/include -I$R/canopy_amd/csrc "$@" -c $R/canopy_amd/csrc/kernels_ls.hip -o /tmp/kernels_ls_$NAME.o
OBJS=$(ls $R/canopy_amd/build/*.o | grep -v kernels_ls)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fvisibility=hidden $OBJS /tmp/kernels_ls_$NAME.o -ldl -o $R/benchmarks/bin/libcanopy_$NAME.so
echo $R/benchmarks/bin/libcanopy_$NAME.so
