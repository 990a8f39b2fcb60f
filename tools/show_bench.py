#!/usr/bin/env python3
"""Pretty-print bench.py JSON lines (roofline_all legs) -- tools/show_bench.py file..."""
import json
import sys
for f in sys.argv[1:]:
    for line in open(f):
        line = line.strip()
        if not line.startswith("{"):
            continue
        r = json.loads(line)
        print(f, "%.1f M frames/s  %.4f ms/step  n_gpus %d" % (r["value"] / 1e6, r["ms_per_step"], r["n_gpus"]))
        if r.get("cpu_baseline"):
            print("   cpu:", r["cpu_baseline"]["value"], r["cpu_baseline"]["sample"])
        for l in r.get("roofline_all") or []:
            print("   %-62s %8.1f us  %s %8.1f %s frac %.3f %s" % (
                l["name"][:62], l["avg_us"], l["bound"], l["achieved"], l["unit"], l["frac"],
                (" hbm %.0f GB/s" % l["hbm_gbs"]) if "hbm_gbs" in l else ""))
