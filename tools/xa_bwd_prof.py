"""Dev tool: the attention backward kernels (8-wave fused, 4-wave fused, split pair) at the configs[1] shape, 20 cold-cache
launches each (a 1-GiB copy between launches), for rocprofv3 --kernel-trace --stats (the profiler's kernel durations do
not depend on host launch gaps).  usage: bash tools/prof_any.sh xab tools/xa_bwd_prof.py [--B 32 --K 375 --L 128 --ilk -1]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from star_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=32); ap.add_argument("--h", type=int, default=8)
ap.add_argument("--L", type=int, default=128); ap.add_argument("--K", type=int, default=375)
ap.add_argument("--hd", type=int, default=128); ap.add_argument("--ilk", type=int, default=-1)
ap.add_argument("--variants", default="2,3")
a = ap.parse_args()
dev = "cuda:0"
B, h, L, K, hd = a.B, a.h, a.L, a.K, a.hd
q = torch.randn(B, L, h, hd, device=dev).bfloat16().permute(0, 2, 1, 3).requires_grad_()
k = torch.randn(B, K, h, hd, device=dev).bfloat16().permute(0, 2, 1, 3).requires_grad_()
v = torch.randn(B, K, h, hd, device=dev).bfloat16().permute(0, 2, 1, 3).requires_grad_()
bias = torch.randn(B, K, device=dev).requires_grad_()
kl = torch.full((B,), K, dtype=torch.int32, device=dev)
go = torch.randn(B, L, h, hd, device=dev).bfloat16().permute(0, 2, 1, 3)
src = torch.empty(1 << 30, dtype=torch.uint8, device=dev).fill_(1)
dst = torch.empty_like(src)
o, _ = ops.xattn(q, k, v, bias=bias, key_len=kl, ilk_diag=a.ilk)
for var in [int(x) for x in a.variants.split(",")]:
    ops.test_override(ops.OVR_XA_BWD, var)
    for i in range(20):
        dst.copy_(src)
        torch.autograd.grad(o, (q, k, v, bias), go, retain_graph=True)
    torch.cuda.synchronize()
ops.test_override(ops.OVR_XA_BWD, 0)
