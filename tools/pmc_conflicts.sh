#!/bin/bash
# LDS bank-conflict cycles of every kernel the given kbench workloads launch, per kernel name (one rocprofv3 --pmc pass
# per workload).  usage: tools/pmc_conflicts.sh  -> gpurun_out/r03/pmc_conflicts.txt
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03; O=gpurun_out/r03/pmc_conflicts.txt; : > $O
run() {
  tag=$1; shift
  ( cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT &&
    timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_BUSY_CU_CYCLES -d gpurun_out/pmc_cf_$tag -o $tag --output-format csv -- python3 tools/kbench.py --iters 2 "$@" > gpurun_out/pmc_cf_$tag.log 2>&1 ) || { echo "$tag failed" >> $O; return 1; }
  echo "== $tag: kbench $*" >> $O
  python3 - >> $O <<PY
import csv, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open('gpurun_out/pmc_cf_$tag/${tag}_counter_collection.csv')):
    acc[r['Kernel_Name'].split('(')[0][:90]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, c in sorted(acc.items()):
    if not any(s in k for s in ('star', 'xattn', 'gemm', 'cif', 'gather', 'topk', 'scorer', 'select', 'colsum', 'score')):
        continue
    m = {n: sum(v) / len(v) for n, v in c.items()}
    print(f"{k:90s} n={len(next(iter(c.values()))):3d} conflict={m.get('SQ_LDS_BANK_CONFLICT', 0):12.0f} lds_active={m.get('SQ_ACTIVE_INST_LDS', 0):12.0f} busy_cu={m.get('SQ_BUSY_CU_CYCLES', 0):12.0f}")
PY
}
run sfwd scorer_fwd && run sbwd scorer_bwd && run xa1 xattn_fwd xattn_bwd && \
run xa3 xattn_fwd xattn_bwd --B 16 --L 227 --K 227 --ilk 2 && run gath gather select && run agg cifagg scan
echo done >> $O
