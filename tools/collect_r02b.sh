#!/bin/bash
# Refresh of the round-2 bench lines and kernel stats (configs[1..3], rates) after a kernel change: the first two blocks
# of tools/collect_r02.sh without the PMC passes.
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
