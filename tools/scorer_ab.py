"""Dev tool: scorer forward GEMM, 4-wave kernel against the 8-wave LDS-DMA kernel (test override STAR_OVR_SCORER_FWD):
scores and ReLU sign bits compared, launches timed with HIP events cache-cold (a 1-GiB copy between launches) and warm.
usage: python tools/scorer_ab.py [--N 48000] [--d 1024]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from star_amd import ops
from star_amd._lib import lib, ptr, stream_ptr, check

ap = argparse.ArgumentParser()
ap.add_argument("--N", type=int, default=48000)
ap.add_argument("--d", type=int, default=1024)
ap.add_argument("--iters", type=int, default=15)
ap.add_argument("--lib", default=None, help="a diagnostic build (tools/build_abl.sh); results are then not compared")
a = ap.parse_args()
if a.lib:
    from star_amd import _lib
    _lib.LIB_PATH = os.path.abspath(a.lib)
dev = "cuda:0"
torch.manual_seed(0)
N, d = a.N, a.d
x = torch.randn(N, d, device=dev).bfloat16()
w1 = (torch.randn(d, d, device=dev) / d ** 0.5).bfloat16()
b1 = torch.randn(d, device=dev) * 0.1
w2 = torch.randn(d, device=dev) / d ** 0.5
b2 = torch.randn(1, device=dev)
L = lib()
wsb = L.star_scorer_fwd_workspace(N, d)
ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
junk_a = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
junk_b = torch.empty(1 << 30, dtype=torch.uint8, device=dev)


def run(scores, bits):
    check(L.star_scorer_fwd(ptr(x), ptr(w1), ptr(b1), ptr(w2), ptr(b2), N, d, ptr(scores), ptr(bits), ptr(ws), wsb,
                            stream_ptr()), "star_scorer_fwd")


def timed(fn, cold):
    ts = []
    for i in range(a.iters):
        if cold:
            junk_b.copy_(junk_a)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


out = {}
for name, ov in (("8-wave LDS-DMA", 0), ("4-wave register-staged", 1)):
    ops.test_override(ops.OVR_SCORER_FWD, ov)
    s = torch.empty(N, dtype=torch.float32, device=dev)
    bits = torch.zeros(L.star_scorer_relu_bits_bytes(N, d), dtype=torch.uint8, device=dev)
    run(s, bits)
    torch.cuda.synchronize()
    out[name] = (s.clone(), bits.clone())
    f = lambda: run(s, bits)
    c, w = timed(f, True), timed(f, False)
    fl = 2.0 * N * d * (d + 1)
    print(f"{name:24s} cold {c:7.1f} us ({fl / c / 1e6:6.1f} TFLOP/s)   warm {w:7.1f} us   (GEMM + finish kernel)")
ops.test_override(ops.OVR_SCORER_FWD, 0)
if a.lib:
    sys.exit(0)
(s8, b8), (s4, b4) = out["8-wave LDS-DMA"], out["4-wave register-staged"]
ref = (torch.relu(x.float() @ w1.float().t() + b1) @ w2 + b2)
print("max |s4 - s8|", (s4 - s8).abs().max().item(), " max |s4 - ref|", (s4 - ref).abs().max().item(),
      " max |s8 - ref|", (s8 - ref).abs().max().item())
nb = (b4 != b8).sum().item()
print("sign-bit bytes that differ:", nb, "of", b4.numel(), "(sums differ in the last bits only where h is ~0)" if nb else "")
