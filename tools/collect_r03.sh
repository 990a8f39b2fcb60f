#!/bin/bash
# Round-3 evidence run on the GPU box: un-profiled bench lines of every BASELINE.json configuration (0-4) and of the
# other compress rates, rocprofv3 kernel stats of configs[1..3], the PMC passes of the scorer GEMM (separate runs per
# counter group), the streaming latency table, the projection stack and a 2-rank gloo rehearsal of the N > 1 bench
# path on one card.  Everything lands under gpurun_out/r03/; copy what is cited into profiles/.
# usage: bash tools/collect_r03.sh [bench|prof|pmc|misc ...]   (default: all parts)
set -o pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03; O=gpurun_out/r03
parts="$*"; [ -z "$parts" ] && parts="bench prof pmc misc"
for part in $parts; do
case $part in
bench)
  for c in 1 0 2 3 4; do
    timeout -k 10 240 python bench.py --config $c > $O/bench_c$c.json 2> $O/bench_c$c.err || { echo "bench c$c failed"; tail -5 $O/bench_c$c.err; exit 1; }
    echo "c$c: $(python tools/show_bench.py $O/bench_c$c.json 2>/dev/null | head -1)"
  done
  for r in 2 8 12; do
    timeout -k 10 200 python bench.py --rate $r --no-cpu-baseline > $O/bench_rate$r.json 2> $O/bench_rate$r.err || { echo "bench rate $r failed"; exit 1; }
  done
  timeout -k 10 120 python bench.py --steps 20 --warmup 5 > $O/bench_c1_driver_shape.json 2> $O/bench_c1_driver_shape.err || { echo "bench 20/5 failed"; exit 1; }
  echo bench done ;;
prof)
  for c in 1 2 3; do
    bash tools/prof.sh r03c$c --config $c --no-roofline-all > $O/prof_c$c.txt 2>&1 || { echo "prof c$c failed"; exit 1; }
    cp gpurun_out/prof_r03c$c/r03c${c}_kernel_stats.csv $O/kernel_stats_c$c.csv
  done
  echo prof done ;;
pmc)
  bash tools/pmc_scorer.sh > $O/pmc_scorer.txt 2>&1 || { echo "pmc scorer failed"; tail -3 $O/pmc_scorer.txt; exit 1; }
  cp gpurun_out/pmc_scorer_fwd.json $O/pmc_scorer_fwd.json
  bash tools/pmc.sh mf "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT" gemm_nt_kernel -- scorer_fwd > $O/pmc_scorer_mfma.txt 2>&1 || exit 1
  bash tools/pmc.sh mf2 "GRBM_GUI_ACTIVE GRBM_COUNT" gemm_nt_kernel -- scorer_fwd > $O/pmc_scorer_grbm.txt 2>&1 || exit 1
  echo pmc done ;;
misc)
  timeout -k 10 200 python tools/stream_latency.py > $O/stream_latency.txt 2>&1 || { echo "stream latency failed"; exit 1; }
  timeout -k 10 300 python tools/f1_stack.py > $O/f1_stack.txt 2>&1 || { echo "f1 stack failed"; tail -3 $O/f1_stack.txt; exit 1; }
  timeout -k 10 200 python tools/xa_ab.py > $O/xa_ab.txt 2>&1 || { echo "xa_ab failed"; exit 1; }
  timeout -k 10 200 python tools/scorer_ab.py > $O/scorer_ab.txt 2>&1 || { echo "scorer_ab failed"; exit 1; }
  for c in 1 4; do
    STAR_DIST_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 2951$c \
      bench.py --gpus 2 --config $c --steps 10 --warmup 3 --no-cpu-baseline --no-roofline-all > $O/bench_gloo2_c$c.json 2> $O/bench_gloo2_c$c.err || { echo "gloo rehearsal c$c failed"; tail -5 $O/bench_gloo2_c$c.err; exit 1; }
  done
  echo misc done ;;
esac
done
echo all done
