#!/bin/bash
# usage: tools/pmc.sh <tag> "<counters>" <kernel-substring> -- <kbench args>   (one rocprofv3 --pmc pass)
tag=$1; ctrs=$2; kern=$3; shift 4
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 120 rocprofv3 --pmc $ctrs -d gpurun_out/pmc_$tag -o $tag --output-format csv -- python3 ${PMC_PROG:-tools/kbench.py --iters 2} "$@" > gpurun_out/pmc_$tag.log 2>&1
echo exit $?
python3 - <<PY
import csv, collections
rows=list(csv.DictReader(open('gpurun_out/pmc_$tag/${tag}_counter_collection.csv')))
acc=collections.defaultdict(list)
for r in rows:
    if '$kern' in r['Kernel_Name']:
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in sorted(acc.items()):
    print(f"{k:28s} n={len(v):3d} mean={sum(v)/len(v):14.1f}")
PY
