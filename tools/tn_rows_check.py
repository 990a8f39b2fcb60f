"""Dev tool: scorer backward (configs[1] shape, R = 12000 of 48000 frames) with the TN and the NT dW1 GEMM, for
scattered rows (every 4th frame) and for one contiguous run -- which form wants which row pattern.
usage: python tools/tn_rows_check.py   (wrap in tools/prof_any.sh for per-kernel times)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from star_amd import ops
from star_amd.nugget import NuggetScorer
dev = "cuda:0"
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
T, d = 1500, 1024
torch.manual_seed(0)
sc = NuggetScorer(d).to(dev)
cases = [(32, 12, False), (32, 8, False), (8, 4, False), (32, 4, False), (32, 4, True), (64, 4, False), (64, 4, True), (128, 4, False), (128, 4, True), (16, 1, True),
         (32, 1, True), (32, 2, False), (32, 2, True)]
for B, every, run in cases:
    x = torch.randn(B, T, d, device=dev).bfloat16().requires_grad_()
    N = B * T
    rows = (torch.arange(0, N // every, dtype=torch.int32, device=dev) if run
            else torch.arange(0, N, every, dtype=torch.int32, device=dev))
    name = f"N={N} R={rows.numel()} {'one run' if run else 'every %dth' % every}"
    g = torch.zeros(N, device=dev)
    g[rows.long()] = torch.randn(rows.numel(), device=dev)
    g = g.view(B, T)
    for force_nt in (0, 1):
        prev = ops.test_override(ops.OVR_SCORER_NT, 1 if force_nt else 2)
        def step():
            x.grad = None
            s = sc(x).squeeze(-1)
            sc.set_active_rows(rows)
            s.backward(g)
        for _ in range(3): step()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=torch.cuda.current_stream()):
            for _ in range(5): step()
        gr.replay(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); gr.replay(); b.record(); torch.cuda.synchronize()
        print(f"{name:34s} {'NT' if force_nt else 'TN'}: forward + backward {a.elapsed_time(b) / 5 * 1e3:7.1f} us", flush=True)
        ops.test_override(ops.OVR_SCORER_NT, prev)
