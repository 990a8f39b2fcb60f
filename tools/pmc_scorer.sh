#!/bin/bash
# HBM-side traffic of the fused scorer GEMM: three separate rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE |
# TCC_EA0 request counts) over tools/kbench.py scorer_fwd, assembled into gpurun_out/pmc_scorer_fwd.json with the
# gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE counts 128-B requests at 64 B -> doubled).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $ctrs -d gpurun_out/pmcs_$i -o p --output-format csv -- python3 tools/kbench.py --iters 2 scorer_fwd > gpurun_out/pmcs_$i.log 2>&1 || { echo "pass $i failed"; exit 1; }
done
python3 - <<'PY'
import csv, collections, json
acc = collections.defaultdict(list)
for i in (1, 2, 3):
    for r in csv.DictReader(open(f"gpurun_out/pmcs_{i}/p_counter_collection.csv")):
        if "gemm_nt_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
ctr = {k: {"mean": sum(v) / len(v), "launches": len(v)} for k, v in sorted(acc.items())}
N, d = 48000, 1024
rd = ctr["FETCH_SIZE"]["mean"] * 1024 * 2
wr = ctr["WRITE_SIZE"]["mean"] * 1024
doc = {"kernel": "gemm_nt_kernel<EPI_SCORE>", "workload": "configs[1] scorer forward: N=48000 frames, d=1024, bf16",
       "command": "tools/pmc_scorer.sh: rocprofv3 --pmc <ctr> -- python3 tools/kbench.py --iters 2 scorer_fwd (one pass per counter group)",
       "counters": ctr,
       "correction": "gfx950: FETCH_SIZE (KB) counts 128-B read requests at 64 B -> doubled (MI355X_MICROARCH.md, HBM section); WRITE_SIZE (KB) taken as is",
       "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "traffic_bytes_per_launch": rd + wr,
       "algorithmic_bytes_per_launch": N * d * 2 + d * d * 2 + N * 4}
json.dump(doc, open("gpurun_out/pmc_scorer_fwd.json", "w"), indent=1)
print(json.dumps({k: doc[k] for k in ("hbm_read_bytes_per_launch", "hbm_write_bytes_per_launch", "traffic_bytes_per_launch")}))
PY
