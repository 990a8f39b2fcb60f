#!/bin/bash
# dev: per-kernel time of gemm_tn_kernel in diagnostic builds (tools/build_abl.sh tnN -DSTAR_GEMM_ABL=N), R = N = 48000
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/tnabl
for tag in base tn1 tn2 tn3 tn16 tn64; do
  lib=tools/libstar_abl_$tag.so; [ $tag = base ] && lib=star_amd/csrc/libstar_nugget.so
  bash tools/prof_any.sh abl_$tag tools/kbench.py --lib $lib scorer_bwd --B 32 --T 1500 --K 1500 --iters 5 > gpurun_out/tnabl/$tag.txt 2>&1 || { echo "$tag failed"; tail -5 gpurun_out/tnabl/$tag.txt; exit 1; }
  echo "$tag: $(grep gemm_tn gpurun_out/tnabl/$tag.txt)"
done
