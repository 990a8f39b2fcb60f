"""Dev tool: A/B of attention kernel variants (test-only overrides) with HIP events, cache-cold (a 512 MB fill between
launches) and cache-warm.  usage: python tools/xa_ab.py [--B ..]"""
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
B, h, L, K, hd = a.B, a.h, a.L, a.K, a.hd
q = torch.randn(B, L, h, hd, device=dev).bfloat16().permute(0, 2, 1, 3).requires_grad_()
k = torch.randn(B, K, h, hd, device=dev).bfloat16().permute(0, 2, 1, 3).requires_grad_()
v = torch.randn(B, K, h, hd, device=dev).bfloat16().permute(0, 2, 1, 3).requires_grad_()
bias = torch.randn(B, K, device=dev).requires_grad_()
kl = torch.full((B,), K, dtype=torch.int32, device=dev)
go = torch.randn(B, L, h, hd, device=dev).bfloat16().permute(0, 2, 1, 3)
junk = torch.empty(512 << 20, dtype=torch.uint8, device=dev)


def timed(fn, cold, n=12):
    ts = []
    for i in range(n):
        if cold:
            junk.fill_(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


for name, sw in (("fwd ring", [(ops.OVR_XA_FWD, 0)]), ("fwd register-staged", [(ops.OVR_XA_FWD, 1)])):
    for w, val in sw:
        ops.test_override(w, val)
    f = lambda: ops.xattn(q, k, v, bias=bias, key_len=kl, ilk_diag=a.ilk)
    print(f"{name:24s} cold {timed(f, True):7.1f} us   warm {timed(f, False):7.1f} us  (one launch between HIP events)")
ops.test_override(ops.OVR_XA_FWD, 0)
o, _ = ops.xattn(q, k, v, bias=bias, key_len=kl, ilk_diag=a.ilk)
for name, sw in (("bwd fused (8 waves)", [(ops.OVR_XA_BWD, 2)]), ("bwd fused (4 waves)", [(ops.OVR_XA_BWD, 3)]),
                 ("bwd split", [(ops.OVR_XA_BWD, 1)])):
    for w, val in sw:
        ops.test_override(w, val)
    f = lambda: torch.autograd.grad(o, (q, k, v, bias), go, retain_graph=True)
    print(f"{name:24s} cold {timed(f, True):7.1f} us   warm {timed(f, False):7.1f} us")
ops.test_override(ops.OVR_XA_BWD, 0)
