"""Dev tool for SURVEY.md 8f item 1: a stack of nl cross-attention layers (q_attn, c_attn, core, c_proj) at the
configs[1] shape, forward + backward, with the reference's per-layer c_attn against share_encoder_kv() (one projection
of the encoder memory for all layers, dK / dV written in place into one gradient buffer).
usage: python tools/f1_stack.py [--shared-only | --per-layer-only]   (the flags are for rocprofv3 runs)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
dev = "cuda:0"
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
B, K, L, d, h, nl = 32, 375, 128, 1024, 8, 4
torch.manual_seed(0)
mem = torch.randn(B, K, d, device=dev).bfloat16()
x = torch.randn(B, L, d, device=dev).bfloat16()
bias = torch.randn(B, K, device=dev)


def timeit(name, fn, iters=3):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=torch.cuda.current_stream()):
        for _ in range(iters): fn()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) / iters * 1e3
    print(f"{name:52s} {us:8.1f} us", flush=True)
    return us

from star_amd.nugget import NuggetCrossAttention, clear_kv_weight_cache, share_encoder_kv, share_scorer_logits
mods = [NuggetCrossAttention(d, h, layer_idx=i).to(dev, torch.bfloat16) for i in range(nl)]
memg = mem.clone().requires_grad_()
xs = [x.clone().requires_grad_() for _ in range(nl)]
gs = [torch.randn_like(x) for _ in range(nl)]
scg = bias.clone().requires_grad_()

def stack(shared):
    # as in a training step: gradients start from None (optimizer.zero_grad(set_to_none=True)), and the weights have
    # just been updated, so the transposed c_attn copies of share_encoder_kv() are rebuilt (once per step)
    for t in [memg, scg] + xs + [p for m in mods for p in m.parameters()]:
        t.grad = None
    clear_kv_weight_cache()
    logits = share_scorer_logits(scg, nl, h)
    if shared:
        kvs = share_encoder_kv(mods, memg)
        outs = [m(xx, encoder_kv=kv, scorer_logits=lg)[0] for m, xx, kv, lg in zip(mods, xs, kvs, logits)]
    else:
        outs = [m(xx, encoder_hidden_states=memg, scorer_logits=lg)[0] for m, xx, lg in zip(mods, xs, logits)]
    torch.autograd.backward(outs, gs)

ts = tt = float("nan")
if "--shared-only" not in sys.argv:
    ts = timeit("4 cross-attention layers fwd+bwd, c_attn per layer", lambda: stack(False))
if "--per-layer-only" not in sys.argv:
    tt = timeit("4 cross-attention layers fwd+bwd, share_encoder_kv", lambda: stack(True))
print(f"decoder cross-attention stack, forward + backward: {ts:.1f} -> {tt:.1f} us")
if "--per-layer-only" not in sys.argv:
    from star_amd.nugget import defer_projection_weight_grads
    defer_projection_weight_grads(mods)
    td = timeit("  + q_attn / c_proj weight gradients in one launch", lambda: stack(True))
    print(f"with defer_projection_weight_grads: {tt:.1f} -> {td:.1f} us")
