#!/bin/bash
# dev: scorer forward, 4-wave kernel, diagnostic builds (tools/build_abl.sh g4_N -DSTAR_GEMM_ABL=N): HIP-event times
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03
for tag in "$@"; do
  lib=tools/g4abl_$tag.so; [ $tag = base ] && lib=star_amd/csrc/libstar_nugget.so
  echo "== $tag"; timeout -k 10 120 python tools/scorer_ab.py --lib $lib 2>&1 | tail -3
done
