#!/bin/bash
# diagnostic library with parts of the scorer GEMM switched off: tools/libstar_abl_TAG.so (never loaded by the product)
# usage: tools/build_abl.sh TAG -DSTAR_GEMM_ABL=n [more flags]
set -e
TAG=$1; shift
cd "$(dirname "$0")/../star_amd/csrc"
mkdir -p build/stamp
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -fno-slp-vectorize"
/opt/rocm/bin/hipcc $F "$@" -c gemm.hip -o build/stamp/gemm_$TAG.o
OBJS="build/stamp/gemm_$TAG.o"
for f in error select gather harness cif cifagg typeembed gatherln xattn xattn_bwd4; do OBJS="$OBJS build/$f.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/libstar_abl_$TAG.so $OBJS
echo built tools/libstar_abl_$TAG.so
