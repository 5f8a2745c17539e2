#!/bin/bash
# Dev tool: build hydrodl2_amd/csrc/libhbvx_<name>.so with extra hipcc flags for ONE translation unit
# (the other objects come from the regular build), for A/B runs with tools/ab_libs.py.
#   tools/build_variant.sh <name> <unit, e.g. launch_stream> <flags...>
set -e
name=$1; unit=$2; shift 2
C=$(dirname $(readlink -f $0))/../hydrodl2_amd/csrc
V=$(dirname $(readlink -f $0))/../gpurun_out/var   # scratch: never shipped to the GPU box
mkdir -p $V
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC "$@" -c $C/$unit.hip -o $V/${unit}_$name.o
objs=""
for o in $C/build/*.o; do b=$(basename $o .o); if [ "$b" = "$unit" ]; then objs="$objs $V/${unit}_$name.o"; else objs="$objs $o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $C/libhbvx_$name.so $objs
echo built $C/libhbvx_$name.so
