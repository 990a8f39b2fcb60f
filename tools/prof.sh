#!/bin/bash
# usage: tools/prof.sh <tag> [bench args...]  -- rocprofv3 kernel stats of a bench run into gpurun_out/prof_<tag>
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o $tag --output-format csv -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" > gpurun_out/prof_$tag.log 2>&1
echo exit $?
python3 - <<PY
import csv
rows=list(csv.DictReader(open('gpurun_out/prof_$tag/${tag}_kernel_stats.csv')))
for r in rows[:32]:
    print(f"{r['Name'][:70]:70s} n={r['Calls']:>4s} avg={float(r['AverageNs'])/1e3:8.1f} tot%={float(r['Percentage']):5.1f}")
PY
