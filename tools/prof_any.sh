#!/bin/bash
# usage: tools/prof_any.sh <tag> <script.py> [args...]  -- rocprofv3 kernel stats of any script into gpurun_out/prof_<tag>
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o $tag --output-format csv -- python3 "$@" > gpurun_out/prof_$tag.log 2>&1
rc=$?
echo exit $rc
[ $rc -eq 0 ] || exit $rc
python3 - <<PY
import csv
rows=list(csv.DictReader(open('gpurun_out/prof_$tag/${tag}_kernel_stats.csv')))
for r in rows[:40]:
    print(f"{r['Name'][:90]:90s} n={r['Calls']:>4s} avg={float(r['AverageNs'])/1e3:8.1f} tot={float(r['TotalDurationNs'])/1e3:9.1f} {float(r['Percentage']):5.1f}%")
PY
