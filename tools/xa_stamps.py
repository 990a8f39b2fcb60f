"""Dev tool: in-kernel timeline of an attention kernel from the -DSTAR_XA_STAMP diagnostic library
(tools/build_stamp.sh).  usage: python tools/xa_stamps.py fwd|bwd|bwd4 [--B ..]   (bwd4: the experimental 4-wave fused backward)"""
import argparse, ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from star_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "tools", "libstar_stamp" + os.environ.get("STAMP_TAG", "") + ".so")
from star_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("what")
ap.add_argument("--B", type=int, default=32); ap.add_argument("--h", type=int, default=8)
ap.add_argument("--L", type=int, default=128); ap.add_argument("--K", type=int, default=375)
ap.add_argument("--hd", type=int, default=128); ap.add_argument("--ilk", type=int, default=-1)
ap.add_argument("--dense", action="store_true", help="K / V dense per head ([B,h,K,hd] memory) instead of head views")
ap.add_argument("--warm", action="store_true", help="do not evict the caches between launches")
a = ap.parse_args()
dev = "cuda:0"
B, h, L, K, hd = a.B, a.h, a.L, a.K, a.hd
q = torch.randn(B, L, h, hd, device=dev).bfloat16().permute(0, 2, 1, 3).requires_grad_()
if a.dense:
    k = torch.randn(B, h, K, hd, device=dev).bfloat16().requires_grad_()
    v = torch.randn(B, h, K, hd, device=dev).bfloat16().requires_grad_()
else:
    k = torch.randn(B, K, h, hd, device=dev).bfloat16().permute(0, 2, 1, 3).requires_grad_()
    v = torch.randn(B, K, h, hd, device=dev).bfloat16().permute(0, 2, 1, 3).requires_grad_()
bias = torch.randn(B, K, device=dev).requires_grad_()
kl = torch.full((B,), K, dtype=torch.int32, device=dev)
junk = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
L_ = _lib.lib()
L_.star_debug_read_stamps.argtypes = [ctypes.c_void_p]
L_.star_debug_read_stamps4.argtypes = [ctypes.c_void_p]
if a.what == "bwd4":
    ops.test_override(ops.OVR_XA_BWD, 3)          # the experimental 4-wave fused kernel (stamps in xattn_bwd4.hip)
    a.what = "bwd"
    four = True
else:
    four = False
buf = np.zeros(1024 * 16, dtype=np.uint64)
for it in range(2):
    if not a.warm:
        junk.fill_(it)                  # evict K / V from the Infinity Cache
    torch.cuda.synchronize()
    if a.what == "fwd":
        o, _ = ops.xattn(q, k, v, bias=bias, key_len=kl, ilk_diag=a.ilk)
    else:
        o, _ = ops.xattn(q, k, v, bias=bias, key_len=kl, ilk_diag=a.ilk)
        go = torch.randn(B, L, h, hd, device=dev).bfloat16().permute(0, 2, 1, 3)
        if not a.warm:
            junk.fill_(it + 7)
        torch.cuda.synchronize()
        torch.autograd.grad(o, (q, k, v, bias), go)
    torch.cuda.synchronize()
    (L_.star_debug_read_stamps4 if four else L_.star_debug_read_stamps)(buf.ctypes.data)
    n = min(B * h * ((L + 127) // 128), 1024)
    st = buf.reshape(1024, 16)[:n].astype(np.int64)
    t0 = st[:, 0].min()
    rel = (st - t0) * 10.0 / 1000.0     # s_memtime ticks at 100 MHz -> us
    print(f"-- launch {it}: per-slot us after the earliest workgroup start: median [min .. max]")
    if a.what == "fwd" and (st[:, 12] != 0).any():
        cyc = (st[:, 12] - st[:, 7]).astype(np.float64)
        us = (st[:, 15] - st[:, 0]) * 0.01
        print(f"   shader clock over the kernel: median {np.median(cyc / us) / 1e3:.2f} GHz")
    for s in range(16):
        if a.what == "fwd" and s in (7, 12):
            continue
        if (st[:, s] == 0).all():
            continue
        col = rel[:, s]
        print(f"   slot {s:2d}: {np.median(col):7.2f} [{col.min():7.2f} .. {col.max():7.2f}]")
    buf[:] = 0
