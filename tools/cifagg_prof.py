"""Dev tool: a few eager CIF-aggregation forward/backward calls at the configs[1] shape (for rocprofv3 --kernel-trace)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from star_amd import ops
B, T, d, L = 32, 1500, 1024, 128
dev = "cuda:0"
torch.manual_seed(0)
al = (torch.rand(B, T, device=dev) * (2.0 * L / T)).requires_grad_()
s = torch.randn(B, T, d, device=dev).bfloat16().requires_grad_()
for _ in range(5):
    out = ops.cif_aggregate(al, s, cap=2 * L + 8)[0]
    torch.autograd.grad(out, (al, s), torch.randn_like(out))
torch.cuda.synchronize()
print("fires", int(ops.cif_aggregate(al, s, cap=2 * L + 8)[4].item()))
