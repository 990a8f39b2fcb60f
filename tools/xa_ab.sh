#!/bin/bash
# A/B of the fused vs split attention backward on a bench shape: tools/xa_ab.sh <bench args...>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for m in fused split; do
  STAR_XA_BWD=$m timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/ab_$m -o ab --output-format csv -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" > gpurun_out/ab_$m.log 2>&1 || exit 1
  tail -1 gpurun_out/ab_$m.log | cut -c1-260
  python3 - <<PY
import csv
for r in csv.DictReader(open('gpurun_out/ab_$m/ab_kernel_stats.csv')):
    if 'xattn' in r['Name']: print('$m', r['Name'][:60], float(r['AverageNs'])/1e3)
PY
done
