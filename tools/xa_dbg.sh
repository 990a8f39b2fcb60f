#!/bin/bash
# timing of the fused attention backward with parts disabled (STAR_XA_DBG bits: 1 no dK/dV stores, 2 no phase C, 4 no slice math, 8 prefetch from group 0)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for d in "$@"; do
  STAR_XA_DBG=$d timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/dbg_$d -o dbg --output-format csv -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/dbg_$d.log 2>&1 || exit 1
  python3 - <<PY
import csv
for r in csv.DictReader(open('gpurun_out/dbg_$d/dbg_kernel_stats.csv')):
    if 'fused_small' in r['Name'] or 'dkv_small' in r['Name'] or 'dq_ring' in r['Name']: print('dbg=$d', r['Name'][:40], float(r['AverageNs'])/1e3)
PY
done
