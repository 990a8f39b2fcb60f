"""Dev tool for the section-8 design note: what the cross-attention projections around the core
cost today at the configs[1] shape -- c_attn (memory -> K,V), q_attn, c_proj on hipBLASLt through torch, the core
forward / backward, and the bytes each moves -- to size what fusing c_attn into the core's K/V staging could save.
usage: python tools/f1_microbench.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from star_amd import ops
dev = "cuda:0"
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
B, K, L, d, h = 32, 375, 128, 1024, 8
hd = d // h
torch.manual_seed(0)
mem = torch.randn(B, K, d, device=dev).bfloat16()
x = torch.randn(B, L, d, device=dev).bfloat16()
w_c = torch.randn(d, 2 * d, device=dev).bfloat16() * 0.02; b_c = torch.zeros(2 * d, device=dev).bfloat16()
w_q = torch.randn(d, d, device=dev).bfloat16() * 0.02; b_q = torch.zeros(d, device=dev).bfloat16()
w_p = torch.randn(d, d, device=dev).bfloat16() * 0.02; b_p = torch.zeros(d, device=dev).bfloat16()
bias = torch.randn(B, K, device=dev)
kl = torch.full((B,), K, dtype=torch.int32, device=dev)

def timeit(name, fn, flops=None, bytes_=None, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=torch.cuda.current_stream()):
        for _ in range(iters): fn()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) / iters * 1e3
    extra = ""
    if flops: extra += f"  {flops / us / 1e6:7.1f} TFLOP/s"
    if bytes_: extra += f"  {bytes_ / 1e6:6.1f} MB  {bytes_ / us / 1e3:7.1f} GB/s"
    print(f"{name:52s} {us:8.1f} us{extra}", flush=True)
    return us

with torch.no_grad():
    kv = torch.addmm(b_c, mem.view(-1, d), w_c)
    t_c = timeit("c_attn: mem[B*K,d] x W[d,2d] (hipBLASLt)", lambda: torch.addmm(b_c, mem.view(-1, d), w_c),
                 flops=2.0 * B * K * d * 2 * d, bytes_=2.0 * (B * K * d + 2 * d * d + B * K * 2 * d))
    t_q = timeit("q_attn: x[B*L,d] x W[d,d]", lambda: torch.addmm(b_q, x.view(-1, d), w_q),
                 flops=2.0 * B * L * d * d, bytes_=2.0 * (2 * B * L * d + d * d))
    t_p = timeit("c_proj: o[B*L,d] x W[d,d]", lambda: torch.addmm(b_p, x.view(-1, d), w_p),
                 flops=2.0 * B * L * d * d, bytes_=2.0 * (2 * B * L * d + d * d))
    q = torch.addmm(b_q, x.view(-1, d), w_q).view(B, L, h, hd).permute(0, 2, 1, 3)
    k = kv.view(B, K, 2, h, hd)[:, :, 0].permute(0, 2, 1, 3)
    v = kv.view(B, K, 2, h, hd)[:, :, 1].permute(0, 2, 1, 3)
    t_f = timeit("core forward (K,V from the c_attn output)", lambda: ops.xattn(q, k, v, bias=bias, key_len=kl),
                 flops=4.0 * B * h * L * K * hd, bytes_=2.0 * (2 * B * L * d + 2 * B * K * d))
print(f"forward today: c_attn {t_c:.1f} + q_attn {t_q:.1f} + core {t_f:.1f} + c_proj {t_p:.1f} = {t_c + t_q + t_f + t_p:.1f} us per layer")
print(f"K/V round trip through HBM: {2 * 2.0 * B * K * d * 2 / 1e6:.1f} MB written by c_attn and read by the core; "
      f"the memory itself is {2.0 * B * K * d / 1e6:.1f} MB")

# ---- step (1) of the design note: ONE c_attn GEMM for all decoder layers (the memory is the same for every layer)
nl = 4
with torch.no_grad():
    w_all = torch.randn(d, nl * 2 * d, device=dev).bfloat16() * 0.02
    b_all = torch.zeros(nl * 2 * d, device=dev).bfloat16()
    ws = [w_all[:, i * 2 * d:(i + 1) * 2 * d].contiguous() for i in range(nl)]
    bs = [b_all[i * 2 * d:(i + 1) * 2 * d].contiguous() for i in range(nl)]
    M = B * K
    fl = 2.0 * M * d * nl * 2 * d
    by = 2.0 * (M * d + nl * 2 * d * d + M * nl * 2 * d)
    t4 = timeit("c_attn x 4 layers, one launch each", lambda: [torch.addmm(bs[i], mem.view(-1, d), ws[i]) for i in range(nl)],
                flops=fl, bytes_=by + 3 * 2.0 * M * d)
    t1 = timeit("c_attn, all layers in one GEMM (N = 8192)", lambda: torch.addmm(b_all, mem.view(-1, d), w_all),
                flops=fl, bytes_=by)
    tcat = timeit("torch.cat of the 4 weights + biases", lambda: (torch.cat(ws, 1), torch.cat(bs, 0)))
    dkv = torch.randn(M, nl * 2 * d, device=dev).bfloat16()
    dkvs = [dkv[:, i * 2 * d:(i + 1) * 2 * d].contiguous() for i in range(nl)]
    m2 = mem.view(-1, d)
    tb4 = timeit("backward x 4: dmem += dKV W^T, dW = mem^T dKV",
                 lambda: [(dkvs[i] @ ws[i].t(), m2.t() @ dkvs[i]) for i in range(nl)], flops=2 * fl)
    tb1 = timeit("backward, all layers: dmem = dKV W^T, dW = mem^T dKV", lambda: (dkv @ w_all.t(), m2.t() @ dkv),
                 flops=2 * fl)
print(f"all-layer c_attn: forward {t4:.1f} -> {t1:.1f} (+ cat {tcat:.1f}) us, backward {tb4:.1f} -> {tb1:.1f} us per step")
