import sys; sys.path.insert(0, '.')
import torch
from star_amd import ops
dev='cuda:0'
for B,T in ((32,1500),(16,3000)):
    torch.manual_seed(0)
    al = torch.sigmoid(torch.randn(B,T,device=dev)-2.5)
    ln = torch.full((B,),T,dtype=torch.int32,device=dev)
    f = lambda: ops.cif_fire_scan(al, lengths=ln, cap=T//4)
    for _ in range(5): f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        f(); torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(10): f()
    g.replay(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    print(f"fire scan B{B} T{T}: {e0.elapsed_time(e1)*100:.1f} us per call (graph of 10, includes its small torch ops)")
