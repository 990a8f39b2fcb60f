#!/bin/bash
# PMC passes of the attention kernels at the configs[1] shape (four separate rocprofv3 --pmc runs per kernel, as the
# counter-block limits require), assembled into gpurun_out/r03/pmc_xattn_{fwd,bwd}.json.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03; O=gpurun_out/r03
for w in fwd bwd; do
  k=xattn_fwd_ring_kernel; [ $w = bwd ] && k=xattn_bwd_fused_small_kernel
  bash tools/pmc.sh xa_${w}_1 "FETCH_SIZE" $k -- xattn_$w > $O/pmc_xattn_${w}_1.txt 2>&1 || exit 1
  bash tools/pmc.sh xa_${w}_2 "WRITE_SIZE" $k -- xattn_$w > $O/pmc_xattn_${w}_2.txt 2>&1 || exit 1
  bash tools/pmc.sh xa_${w}_3 "SQ_WAIT_INST_ANY SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" $k -- xattn_$w > $O/pmc_xattn_${w}_3.txt 2>&1 || exit 1
  bash tools/pmc.sh xa_${w}_4 "GRBM_GUI_ACTIVE GRBM_COUNT" $k -- xattn_$w > $O/pmc_xattn_${w}_4.txt 2>&1 || exit 1
  python3 - <<PY
import json, re
c = {}
for i in (1, 2, 3, 4):
    for l in open("$O/pmc_xattn_${w}_%d.txt" % i):
        m = re.match(r"(\w+)\s+n=\s*(\d+)\s+mean=\s*([\d.]+)", l)
        if m:
            c[m.group(1)] = float(m.group(3))
B, h, L, K, hd = 32, 8, 128, 375, 128
q = B * h * L * hd * 2; kv = B * h * K * hd * 2
alg = {"fwd": {"read": q + 2 * kv + B * K * 4, "write": q + B * h * L * 4},
       "bwd": {"read": 3 * q + 2 * kv + B * h * L * 4 + B * K * 4, "write": q + 2 * kv + B * h * K * 4}}["$w"]
rd, wr = c["FETCH_SIZE"] * 1024 * 2, c["WRITE_SIZE"] * 1024
doc = {"kernel": "$k", "workload": "configs[1] attention core, one layer: B=32, h=8, L=128, K=375, hd=128, bf16, score bias on",
       "command": "tools/pmc_xattn.sh: tools/pmc.sh xa_${w}_N \"<counters>\" <kernel> -- xattn_$w (four separate rocprofv3 --pmc passes over tools/kbench.py --iters 2; back-to-back launches, so K/V may be Infinity-Cache warm)",
       "counters": c,
       "correction": "gfx950: FETCH_SIZE (KB) counts 128-B read requests at 64 B -> doubled (MI355X_MICROARCH.md, HBM section); WRITE_SIZE (KB) as is; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_BUSY_CU_CYCLES count in units of 4 cycles",
       "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "algorithmic_bytes_per_launch": alg,
       "derived": {"traffic_over_algorithmic": (rd + wr) / (alg["read"] + alg["write"]),
                   "wait_inst_any_fraction_of_wave_cycles": c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"],
                   "mfma_busy_cycles_per_simd": c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024,
                   "mfma_utilisation_of_wave_lifetime": c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (c["SQ_WAVE_CYCLES"] * 4 / c["SQ_WAVES"]),
                   "lds_bank_conflict_cycles_per_cu": c["SQ_LDS_BANK_CONFLICT"] / 256}}
json.dump(doc, open("$O/pmc_xattn_$w.json", "w"), indent=1)
print("$w", json.dumps(doc["derived"]))
PY
done
