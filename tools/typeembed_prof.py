"""Eager timing loop for the type-embedding add (fwd + bwd) at the bench shape; run under rocprofv3 --kernel-trace."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from star_amd import ops

B, T, d = 32, 1500, 1024
dev = "cuda:0"
x = torch.randn(B, T, d, device=dev).to(torch.bfloat16).requires_grad_()
w = (torch.randn(2, d, device=dev) * 0.1).to(torch.bfloat16).requires_grad_()
table = (torch.rand(B, T, device=dev) < 0.25).long()
g = torch.randn(B, T, d, device=dev).to(torch.bfloat16)
for it in range(12):
    out = ops.type_embed_add(x, table, w)
    out.backward(g)
    # what the reference line launches: embedding lookup + add, and their backward
    x2, w2 = x.detach().requires_grad_(), w.detach().requires_grad_()
    ref = x2 + torch.nn.functional.embedding(table, w2)
    ref.backward(g)
torch.cuda.synchronize()
print("ok")
