"""Dev tool: list, per kernel of hipcc -save-temps assembly files, the s_waitcnt vmcnt(0) that sit inside loops (the\nusual sign of a drained prefetch / LDS-DMA pipeline).  usage: python tools/isa_scan.py <file.s> ..."""
import re,sys
for f in sys.argv[1:]:
    lines=open(f).read().split('\n')
    # kernel ranges
    kern=None; start=0
    kernels=[]
    for i,l in enumerate(lines):
        m=re.match(r'^(_ZN4star\w+|_Z\w+):',l)
        if m: kern=m.group(1); start=i
        if 's_endpgm' in l and kern: kernels.append((kern,start,i)); kern=None
    for k,a,b in kernels:
        labels={}
        for i in range(a,b):
            m=re.match(r'^(\.LBB\d+_\d+):',lines[i])
            if m: labels[m.group(1)]=i
        loops=[]
        for i in range(a,b):
            m=re.search(r's_cbranch_\w+\s+(\.LBB\d+_\d+)',lines[i]) or re.search(r's_branch\s+(\.LBB\d+_\d+)',lines[i])
            if m and m.group(1) in labels and labels[m.group(1)]<i: loops.append((labels[m.group(1)],i))
        hits=set()
        for la,lb in loops:
            for i in range(la,lb):
                if re.search(r's_waitcnt.*vmcnt\(0\)',lines[i]): hits.add(i)
        if hits:
            print(f"{f.split('-')[0]} {k[:70]}: {len(hits)} vmcnt(0) inside loops, lines {sorted(h-a for h in hits)[:12]}")
