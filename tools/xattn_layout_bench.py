"""Dev tool: attention forward / backward on K/V given as [B,K,h,hd]-ordered views (what split_heads yields) vs dense
[B,h,K,hd] tensors, same sizes (configs[1]).  Run under rocprofv3 --kernel-trace --stats and compare kernel times."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from star_amd import ops
B, h, L, K, hd = 32, 8, 128, 375, 128
dev = "cuda:0"
mode = sys.argv[1] if len(sys.argv) > 1 else "view"
q = torch.randn(B, L, h, hd, device=dev).bfloat16().permute(0, 2, 1, 3).requires_grad_()
if mode == "view":
    k = torch.randn(B, K, h, hd, device=dev).bfloat16().permute(0, 2, 1, 3).requires_grad_()
    v = torch.randn(B, K, h, hd, device=dev).bfloat16().permute(0, 2, 1, 3).requires_grad_()
else:
    k = torch.randn(B, h, K, hd, device=dev).bfloat16().requires_grad_()
    v = torch.randn(B, h, K, hd, device=dev).bfloat16().requires_grad_()
sc = torch.randn(B, K, device=dev).requires_grad_()
for _ in range(12):
    o, _ = ops.xattn(q, k, v, bias=sc)
    o.backward(torch.ones_like(o))
torch.cuda.synchronize()
print("ok", mode)
