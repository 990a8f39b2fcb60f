#!/bin/bash
# Round-2 evidence run on the GPU box: un-profiled bench lines of every BASELINE.json configuration, rocprofv3 kernel
# stats of configs[1..3], and PMC passes (separate runs per counter group) for the scorer GEMM and the attention
# kernels.  Everything lands under gpurun_out/r02/; copy what is cited into profiles/.
set -o pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02; O=gpurun_out/r02
for c in 1 2 3; do
  timeout -k 10 200 python bench.py --config $c > $O/bench_c$c.json 2> $O/bench_c$c.err || { echo "bench c$c failed"; exit 1; }
done
for r in 2 8 12; do
  timeout -k 10 200 python bench.py --rate $r --no-cpu-baseline > $O/bench_rate$r.json 2> $O/bench_rate$r.err || { echo "bench rate $r failed"; exit 1; }
done
echo bench done
for c in 1 2 3; do
  bash tools/prof.sh r02c$c --config $c --no-roofline-all > $O/prof_c$c.txt 2>&1 || { echo "prof c$c failed"; exit 1; }
  cp gpurun_out/prof_r02c$c/r02c${c}_kernel_stats.csv $O/kernel_stats_c$c.csv
done
echo prof done
bash tools/pmc_scorer.sh > $O/pmc_scorer.txt 2>&1 || { echo "pmc scorer failed"; exit 1; }
cp gpurun_out/pmc_scorer_fwd.json $O/pmc_scorer_fwd.json
bash tools/pmc.sh mf "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT" gemm_nt_kernel -- scorer_fwd > $O/pmc_scorer_mfma.txt 2>&1 || exit 1
bash tools/pmc.sh mf2 "GRBM_GUI_ACTIVE GRBM_COUNT" gemm_nt_kernel -- scorer_fwd > $O/pmc_scorer_grbm.txt 2>&1 || exit 1
echo pmc scorer done
for w in fwd bwd; do
  k=xattn_fwd_ring_kernel; [ $w = bwd ] && k=xattn_bwd_fused_small_kernel
  bash tools/pmc.sh xa_${w}_1 "FETCH_SIZE" $k -- xattn_$w > $O/pmc_xattn_${w}_fetch.txt 2>&1 || exit 1
  bash tools/pmc.sh xa_${w}_2 "WRITE_SIZE" $k -- xattn_$w > $O/pmc_xattn_${w}_write.txt 2>&1 || exit 1
  bash tools/pmc.sh xa_${w}_3 "SQ_WAIT_INST_ANY SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" $k -- xattn_$w > $O/pmc_xattn_${w}_sq.txt 2>&1 || exit 1
  bash tools/pmc.sh xa_${w}_4 "GRBM_GUI_ACTIVE GRBM_COUNT" $k -- xattn_$w > $O/pmc_xattn_${w}_grbm.txt 2>&1 || exit 1
done
echo all done
