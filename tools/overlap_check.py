"""Dev tool: would the row-gather backward (HBM-bound) hide beside the scorer backward (MFMA / LDS-bound)?  Both
captured in one HIP graph, once one after the other on one stream, once forked onto two streams.
usage: python tools/overlap_check.py [--B 32 --T 1500 --K 375]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from star_amd import ops
from star_amd.nugget import NuggetScorer
ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=32); ap.add_argument("--T", type=int, default=1500)
ap.add_argument("--K", type=int, default=375); ap.add_argument("--d", type=int, default=1024)
a = ap.parse_args()
dev = "cuda:0"
B, T, K, d = a.B, a.T, a.K, a.d
main = torch.cuda.Stream(); side = torch.cuda.Stream()
torch.cuda.set_stream(main)
torch.manual_seed(0)
sc = NuggetScorer(d).to(dev)
x = torch.randn(B, T, d, device=dev).bfloat16().requires_grad_()
s = sc(x).squeeze(-1)
idx = torch.stack([torch.randperm(T, device=dev)[:K].sort().values for _ in range(B)])
rows = (idx + torch.arange(B, device=dev)[:, None] * T).reshape(-1).int()
gs = torch.zeros(B * T, device=dev); gs[rows.long()] = 1.0; gs = gs.view(B, T)
src = torch.randn(B, T, d, device=dev).bfloat16().requires_grad_()
with torch.cuda.stream(side):
    side.wait_stream(main)
    mem = ops.gather_rows(src, idx)            # forward on the side stream: autograd runs its backward there
main.wait_stream(side)
mem_main = ops.gather_rows(src, idx)           # same op with its forward on the main stream
gm = torch.randn_like(mem)
params = list(sc.parameters())

def scorer_bwd():
    sc.set_active_rows(rows)
    return torch.autograd.grad(s, [x] + params, gs, retain_graph=True)

def serial():
    scorer_bwd()
    torch.autograd.grad(mem_main, src, gm, retain_graph=True)

def forked():
    side.wait_stream(main)
    with torch.cuda.stream(side):
        torch.autograd.grad(mem, src, gm, retain_graph=True)
    scorer_bwd()
    main.wait_stream(side)

def timeit(name, fn, iters=5):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=main):
        for _ in range(iters): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    print(f"{name:44s} {e0.elapsed_time(e1) / iters * 1e3:8.1f} us", flush=True)

timeit("scorer backward alone", scorer_bwd)
timeit("gather backward alone", lambda: torch.autograd.grad(mem_main, src, gm, retain_graph=True))
timeit("scorer backward, then gather backward", serial)
timeit("gather backward on a side stream beside it", forked)
