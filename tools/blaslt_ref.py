"""Dev tool: what torch's bf16 GEMM (hipBLASLt) reaches on the scorer's shape, [48000, 1024] x [1024, 1024]^T, with and
without materialising H -- the library-class reference point for gemm_nt_kernel<EPI_SCORE>.  usage: python tools/blaslt_ref.py"""
import torch
dev = "cuda:0"
torch.manual_seed(0)
N, d = 48000, 1024
x = torch.randn(N, d, device=dev).bfloat16()
w1 = (torch.randn(d, d, device=dev) / d ** 0.5).bfloat16()
b1 = (torch.randn(d, device=dev) * 0.1).bfloat16()
w2 = (torch.randn(d, 1, device=dev) / d ** 0.5).bfloat16()
junk_a = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
junk_b = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
out = torch.empty(N, d, device=dev, dtype=torch.bfloat16)


def timed(fn, cold, n=15):
    ts = []
    for _ in range(n):
        if cold:
            junk_b.copy_(junk_a)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


fl = 2.0 * N * d * d
for name, fn in (("H = x @ W1^T (bf16 out, 98 MB written)", lambda: torch.mm(x, w1.t(), out=out)),
                 ("H = addmm(b1, x, W1^T)", lambda: torch.addmm(b1, x, w1.t(), out=out)),
                 ("scores = relu(addmm(b1, x, W1^T)) @ w2 (the reference's op sequence, bf16)",
                  lambda: torch.relu_(torch.addmm(b1, x, w1.t(), out=out)) @ w2)):
    for _ in range(3):
        fn()
    c, w = timed(fn, True), timed(fn, False)
    print(f"{name:80s} cold {c:7.1f} us ({fl / c / 1e6:6.0f} TFLOP/s)   warm {w:7.1f} us ({fl / w / 1e6:6.0f} TFLOP/s)")
