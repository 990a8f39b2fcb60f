#!/bin/bash
# rocprofv3 kernel stats of the shared-K/V cross-attention stack (tools/f1_stack.py --shared-only): what the 1.55 ms are.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_f1 -o f1 --output-format csv -- python3 tools/f1_stack.py --shared-only > gpurun_out/prof_f1.log 2>&1
echo exit $?
python3 - <<PY
import csv
rows=list(csv.DictReader(open('gpurun_out/prof_f1/f1_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
# 3 warm-up calls + 2 replays x 3 = 9 stack passes (the capture itself runs nothing)
for r in rows[:40]:
    print(f"{r['Name'][:100]:100s} n={r['Calls']:>4s} avg={float(r['AverageNs'])/1e3:8.1f} per_stack={float(r["TotalDurationNs"])/9e3:8.1f} tot%={float(r['Percentage']):5.1f}")
print('sum per stack us', tot/9e3)
PY
