#!/bin/bash
# diagnostic library with in-kernel time stamps: tools/libstar_stamp[_TAG].so (never loaded by the product path)
# usage: tools/build_stamp.sh [TAG [extra -D flags...]]
set -e
TAG=$1; shift || true
cd "$(dirname "$0")/../star_amd/csrc"
mkdir -p build/stamp
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value"
/opt/rocm/bin/hipcc $F -mllvm -amdgpu-mfma-vgpr-form=1 -DSTAR_XA_STAMP "$@" -c xattn.hip -o build/stamp/xattn$TAG.o
/opt/rocm/bin/hipcc $F -DSTAR_XA_STAMP "$@" -c xattn_bwd4.hip -o build/stamp/xattn_bwd4$TAG.o
OBJS="build/stamp/xattn$TAG.o build/stamp/xattn_bwd4$TAG.o"
for f in error select gather harness cif cifagg typeembed gatherln gemm; do OBJS="$OBJS build/$f.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/libstar_stamp$TAG.so $OBJS
echo built tools/libstar_stamp$TAG.so
