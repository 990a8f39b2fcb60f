"""Dev tool: the row-gather backward issues its row loads as inline asm (untracked by the compiler) and waits for them by
hand.  That is only sound while no instruction touches a destination register between its load and the wait -- checked
here on the compiler's assembly.  usage: hipcc ... -save-temps -c star_amd/csrc/gather.hip ; python tools/asm_load_check.py
<gather-hip-amdgcn-amd-amdhsa-gfx950.s>"""
import re, sys
lines = open(sys.argv[1]).read().split('\n')
start = None
bad_total = 0
for i, l in enumerate(lines):
    m = re.match(r'^(_ZN4star22gather_rows_bwd_kernel\w+):', l)
    if m:
        start, name = i, m.group(1)
    if start is not None and 's_endpgm' in l:
        body = lines[start:i]
        inapp, loads = False, []
        for k, b in enumerate(body):
            if 'ASMSTART' in b: inapp = True
            if 'ASMEND' in b: inapp = False
            mm = re.search(r'global_load_dwordx4 (v\[\d+:\d+\])', b)
            if mm and inapp: loads.append((k, mm.group(1)))
        bad = []
        for li, r in loads:
            wait = next(k for k in range(li, len(body)) if 's_waitcnt vmcnt(0)' in body[k])
            a, z = map(int, re.findall(r'\d+', r))
            regs = set(range(a, z + 1))
            for k in range(li + 1, wait):
                if body[k].strip().startswith(';'): continue
                for mm in re.finditer(r'v\[(\d+):(\d+)\]|\bv(\d+)\b', body[k]):
                    rr = set(range(int(mm.group(1)), int(mm.group(2)) + 1)) if mm.group(1) else {int(mm.group(3))}
                    if rr & regs: bad.append((r, body[k].strip()))
        print(f"{name[:60]}: {len(loads)} asm loads, {len(bad)} touches of a destination before its wait")
        bad_total += len(bad)
        start = None
sys.exit(1 if bad_total else 0)
