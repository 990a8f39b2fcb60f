"""Dev tool: time individual C-ABI kernels with HIP events (graph-replayed so host overhead is excluded).
usage: python tools/kbench.py [--lib path.so] xattn_fwd|xattn_bwd|scorer_fwd|scorer_bwd|select|gather ..."""
import argparse, ctypes, os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from star_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=None)
ap.add_argument("what", nargs="+")
ap.add_argument("--B", type=int, default=32); ap.add_argument("--T", type=int, default=1500)
ap.add_argument("--d", type=int, default=1024); ap.add_argument("--h", type=int, default=8)
ap.add_argument("--L", type=int, default=128); ap.add_argument("--K", type=int, default=375)
ap.add_argument("--ilk", type=int, default=-1); ap.add_argument("--iters", type=int, default=20)
a = ap.parse_args()
if a.lib:
    _lib.LIB_PATH = os.path.abspath(a.lib)
from star_amd import ops
dev = "cuda:0"
torch.manual_seed(0)
# everything runs on one non-default stream: a graph capture needs one, and autograd runs a backward node on the stream
# its forward ran on, so forwards built outside the capture must live on the capture stream too
_stream = torch.cuda.Stream()
torch.cuda.set_stream(_stream)

def timeit(name, fn, flops=None, bytes_=None):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=torch.cuda.current_stream()):
        for _ in range(a.iters): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / a.iters * 1e3
    extra = ""
    if flops: extra += f"  {flops/us/1e6:8.1f} TFLOP/s"
    if bytes_: extra += f"  {bytes_/us/1e3:8.1f} GB/s"
    print(f"{name:28s} {us:9.2f} us{extra}", flush=True)

B, T, d, h, L, K = a.B, a.T, a.d, a.h, a.L, a.K
hd = d // h
for w in a.what:
    if w in ("xattn_fwd", "xattn_bwd"):
        q = torch.randn(B, L, h, hd, device=dev).bfloat16().permute(0, 2, 1, 3).requires_grad_()
        k = torch.randn(B, K, h, hd, device=dev).bfloat16().permute(0, 2, 1, 3).requires_grad_()
        v = torch.randn(B, K, h, hd, device=dev).bfloat16().permute(0, 2, 1, 3).requires_grad_()
        bias = torch.randn(B, K, device=dev).requires_grad_()
        kl = torch.full((B,), K, dtype=torch.int32, device=dev)
        fl = 4.0 * B * h * L * K * hd
        if w == "xattn_fwd":
            with torch.no_grad():
                timeit("xattn_fwd", lambda: ops.xattn(q, k, v, bias=bias, key_len=kl, ilk_diag=a.ilk), flops=fl)
        else:
            o, _ = ops.xattn(q, k, v, bias=bias, key_len=kl, ilk_diag=a.ilk)
            go = torch.randn(B, L, h, hd, device=dev).bfloat16().permute(0, 2, 1, 3)
            timeit("xattn_bwd", lambda: torch.autograd.grad(o, (q, k, v, bias), go, retain_graph=True), flops=2.5 * fl)
    elif w in ("scorer_fwd", "scorer_bwd"):
        from star_amd.nugget import NuggetScorer
        sc = NuggetScorer(d).to(dev)
        x = torch.randn(B, T, d, device=dev).bfloat16().requires_grad_()
        N = B * T
        if w == "scorer_fwd":
            with torch.no_grad():
                timeit("scorer_fwd (all launches)", lambda: sc(x), flops=N * (2.0 * d * d + 2 * d))
        else:
            s = sc(x).squeeze(-1)
            rows = torch.stack([torch.randperm(T, device=dev)[:K] + b * T for b in range(B)]).reshape(-1).int()
            gs = torch.zeros_like(s); gs.view(-1)[rows.long()] = 1.0
            R = rows.numel()

            def bwd():
                sc.set_active_rows(rows)          # single-use hint
                return torch.autograd.grad(s, (x, *sc.parameters()), gs, retain_graph=True)
            timeit("scorer_bwd (+dX)", bwd, flops=R * 6.0 * d * d)
    elif w == "select":
        s = torch.randn(B, T, device=dev)
        ln = torch.full((B,), T, dtype=torch.int32, device=dev)
        timeit("topk_select", lambda: ops.topk_select(s, lengths=ln, rate=4, K=K))
    elif w == "gather":
        src = torch.randn(B, T, d, device=dev).bfloat16().requires_grad_()
        idx = torch.stack([torch.randperm(T, device=dev)[:K].sort().values for _ in range(B)])
        timeit("gather_fwd", lambda: ops.gather_rows(src.detach(), idx), bytes_=2.0 * B * K * d * 2)
        m = ops.gather_rows(src, idx); gm = torch.randn_like(m)
        timeit("gather_bwd", lambda: torch.autograd.grad(m, src, gm, retain_graph=True), bytes_=(B * K + B * T) * d * 2.0)
    elif w == "scan":
        al = torch.rand(B, T, device=dev) * 0.15
        ln = torch.full((B,), T, dtype=torch.int32, device=dev)
        tl = torch.full((B,), int(0.075 * T) + 2, dtype=torch.int32, device=dev)
        timeit("cif_resize", lambda: ops.cif_resize(al, tl))
        timeit("cif_fire_scan", lambda: ops.cif_fire_scan(al, lengths=ln, cap=int(0.075 * T) + 4))
    elif w == "cifagg":
        al = (torch.rand(B, T, device=dev) * (2.0 * L / T)).requires_grad_()       # ~L fires per utterance
        s = torch.randn(B, T, d, device=dev).bfloat16().requires_grad_()
        cap = 2 * L + 8
        with torch.no_grad():
            timeit("cif_aggregate fwd", lambda: ops.cif_aggregate(al, s, cap=cap), bytes_=(B * T + B * L) * d * 2.0)
        out = ops.cif_aggregate(al, s, cap=cap)[0]
        go = torch.randn_like(out)
        timeit("cif_aggregate bwd", lambda: torch.autograd.grad(out, (al, s), go, retain_graph=True),
               bytes_=2.0 * B * T * d * 2.0)
