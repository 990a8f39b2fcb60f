"""Dev tool: attention core forward / backward with and without attention-probability dropout (configs[1] shape by
default), graph-replayed.  usage: python tools/drop_bench.py [--B 32 --L 128 --K 375 --ilk -1]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from star_amd import ops
ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=32); ap.add_argument("--h", type=int, default=8)
ap.add_argument("--L", type=int, default=128); ap.add_argument("--K", type=int, default=375)
ap.add_argument("--hd", type=int, default=128); ap.add_argument("--ilk", type=int, default=-1)
a = ap.parse_args()
dev = "cuda:0"
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
B, h, L, K, hd = a.B, a.h, a.L, a.K, a.hd
q = torch.randn(B, L, h, hd, device=dev).bfloat16().permute(0, 2, 1, 3).requires_grad_()
k = torch.randn(B, K, h, hd, device=dev).bfloat16().permute(0, 2, 1, 3).requires_grad_()
v = torch.randn(B, K, h, hd, device=dev).bfloat16().permute(0, 2, 1, 3).requires_grad_()
bias = torch.randn(B, K, device=dev).requires_grad_()
kl = torch.full((B,), K, dtype=torch.int32, device=dev)
go = torch.randn(B, L, h, hd, device=dev).bfloat16().permute(0, 2, 1, 3)

def timeit(name, fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=torch.cuda.current_stream()):
        for _ in range(iters): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    print(f"{name:34s} {e0.elapsed_time(e1) / iters * 1e3:8.1f} us", flush=True)

for p in (0.0, 0.1):
    rng = (1234, 0) if p > 0 else None
    with torch.no_grad():
        timeit(f"forward  p={p}", lambda: ops.xattn(q, k, v, bias=bias, key_len=kl, ilk_diag=a.ilk, dropout_p=p, dropout_rng=rng))
    o, _ = ops.xattn(q, k, v, bias=bias, key_len=kl, ilk_diag=a.ilk, dropout_p=p, dropout_rng=rng)
    timeit(f"backward p={p}", lambda: torch.autograd.grad(o, (q, k, v, bias), go, retain_graph=True))
