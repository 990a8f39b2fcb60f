"""Dev check: torch.autograd.grad through the star ops, eagerly and under HIP-graph capture.
   arg 'torch' runs the capture part with a plain torch op instead (to tell our code from torch's)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from star_amd import ops
dev = "cuda:0"
B, h, L, K, hd = 4, 8, 64, 100, 64
q = torch.randn(B, h, L, hd, device=dev).bfloat16().requires_grad_()
k = torch.randn(B, h, K, hd, device=dev).bfloat16().requires_grad_()
v = torch.randn(B, h, K, hd, device=dev).bfloat16().requires_grad_()
sc = torch.randn(B, K, device=dev).requires_grad_()
plain = len(sys.argv) > 1 and sys.argv[1] == "torch"

def f():
    if plain:
        return torch.softmax(q.float() @ k.float().transpose(-1, -2) + sc[:, None, None, :], -1) @ v.float()
    return ops.xattn(q, k, v, bias=sc)[0]

g = torch.randn(B, h, L, hd, device=dev, dtype=torch.float32 if plain else torch.bfloat16)
if len(sys.argv) > 2:        # an eager pass on the default stream first
    r = torch.autograd.grad(f(), [q, k, v, sc], g)
    print('eager ok', flush=True)
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    for _ in range(2):
        r = torch.autograd.grad(f(), [q, k, v, sc], g)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=st):
        r = torch.autograd.grad(f(), [q, k, v, sc], g)
    gr.replay(); torch.cuda.synchronize()
print("captured autograd.grad ok", "torch" if plain else "star", flush=True)
