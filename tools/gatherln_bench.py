"""Dev tool: ln_f over all frames + row gather (what the reference does, torch kernels) against star gather_layernorm,
forward and backward, at the configs[1] shape.  usage: python tools/gatherln_bench.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from star_amd import ops
dev = "cuda:0"
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
B, T, K, d = 32, 1500, 375, 1024
torch.manual_seed(0)
h = torch.randn(B, T, d, device=dev).bfloat16().requires_grad_()
idx = torch.stack([torch.randperm(T, device=dev)[:K].sort().values for _ in range(B)])
ln = torch.nn.LayerNorm(d).to(dev)
go = torch.randn(B, K, d, device=dev).bfloat16()
ar = torch.arange(B, device=dev).unsqueeze(1)

def timeit(name, fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=torch.cuda.current_stream()):
        for _ in range(iters): fn()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); b.record(); torch.cuda.synchronize()
    print(f"{name:44s} {a.elapsed_time(b) / iters * 1e3:8.1f} us", flush=True)

def ref_fwd():
    return torch.nn.functional.layer_norm(h, (d,), ln.weight.bfloat16(), ln.bias.bfloat16(), 1e-5)[ar, idx]
def star_fwd():
    return ops.gather_layernorm(h, idx, ln.weight, ln.bias, 1e-5)
with torch.no_grad():
    timeit("torch: ln_f over B*T rows, then gather (fwd)", ref_fwd)
    timeit("star gather_layernorm (fwd)", star_fwd)
params = (h, ln.weight, ln.bias)
o1 = ref_fwd(); o2 = star_fwd()
timeit("torch: backward of the same", lambda: torch.autograd.grad(o1, params, go, retain_graph=True))
timeit("star gather_layernorm backward (+ scatter)", lambda: torch.autograd.grad(o2, params, go, retain_graph=True))
