"""Dev tool: gather_rows backward with and without -1 pad slots (duplicates on the last frame), eager HIP-event timing."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from star_amd import _lib
from star_amd.ops import DT_BF16
B, T, K, d = 32, 1500, 116, 1024
if len(sys.argv) > 3:
    B, T, K = (int(v) for v in sys.argv[1:4])
pads = [int(v) for v in sys.argv[4:]] or [0, 1, 2, 8, 58]
dev = "cuda:0"
g = torch.randn(B, K, d, device=dev).bfloat16()
ds = torch.empty(B, T, d, device=dev, dtype=torch.bfloat16)
L = _lib.lib()
for npad in pads:
    idx = torch.stack([torch.randperm(T - 1, device=dev)[:K].sort().values for _ in range(B)])
    if npad:
        idx[:, K - npad:] = -1
    def run():
        _lib.check(L.star_gather_rows_bwd(_lib.ptr(g), _lib.ptr(idx), _lib.ptr(ds), B, T, K, d, DT_BF16, _lib.stream_ptr()), "x")
    for _ in range(3): run()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): run()
    b.record(); torch.cuda.synchronize()
    print(f"pads={npad:3d}: {a.elapsed_time(b) / 20 * 1e3:8.1f} us", flush=True)
