"""Dev tool: timeline of ONE replayed step from a rocprofv3 --kernel-trace csv: start, duration and the idle gap before
every kernel (gap = start - max end so far), so that hand-over gaps inside the graph show up.
usage: python tools/step_timeline.py <kernel_trace.csv> [first-kernel-substring (default cast_transpose)] [which step from the end (default 3)]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
first = sys.argv[2] if len(sys.argv) > 2 else "cast_transpose"
back = int(sys.argv[3]) if len(sys.argv) > 3 else 3
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
starts = [i for i, k in enumerate(ks) if first in k[2]]
a, b = starts[-back], starts[-back + 1]
t0 = ks[a][0]
end = t0
tot_gap = 0.0
for s, e, n in ks[a:b]:
    gap = (s - end) / 1e3
    if gap > 0:
        tot_gap += gap
    print(f"{(s - t0) / 1e3:8.1f} us  +{(e - s) / 1e3:7.1f}  gap {gap:6.1f}  {n[:100]}")
    end = max(end, e)
print(f"step span {(ks[b][0] - t0) / 1e3:.1f} us, idle gaps {tot_gap:.1f} us, last kernel to next step {(ks[b][0] - end) / 1e3:.1f} us")
