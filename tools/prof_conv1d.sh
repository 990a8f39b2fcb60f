#!/bin/bash
# kernel trace of tools/conv1d_forms.py: which kernel every call form lands on and how long it runs (runs of the same
# kernel name in launch order; the flush fills are dropped)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03
timeout -k 10 300 rocprofv3 --kernel-trace -d gpurun_out/prof_c1d -o c1d --output-format csv -- python3 tools/conv1d_forms.py "$@" > gpurun_out/prof_c1d.log 2>&1
echo exit $?
python3 - <<PY
import csv, collections
rows = list(csv.DictReader(open('gpurun_out/prof_c1d/c1d_kernel_trace.csv')))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [l.split('warm')[0].strip() for l in open('gpurun_out/prof_c1d.log') if ' warm ' in l]
form = -1
acc = collections.OrderedDict()
for r in rows:
    n = r['Kernel_Name']
    if 'arange' in n:
        form += 1
        continue
    if form < 0 or "Cijk" not in n and "star::" not in n and "elementwise" not in n and "reduce" not in n or 'FillFunctor<unsigned char>' in n or 'fillBuffer' in n:
        continue
    dur = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    acc.setdefault(form, collections.OrderedDict()).setdefault(n, []).append(dur)
for f, ks in acc.items():
    print(names[f] if f < len(names) else f"form {f}")
    tot = 0.0
    for n, d in ks.items():
        d2 = sorted(d)
        # 26 calls per form: 13 back to back (warm), 13 after a 512 MB fill (cold)
        per_call = len(d) / 26.0
        tot += d2[len(d2) // 2] * per_call
        print(f"    {n[:104]:104s} n={len(d):3d} median={d2[len(d2)//2]:7.1f} min={d2[0]:7.1f}")
    print(f"    -> {tot:7.1f} us of kernels per call")
PY
