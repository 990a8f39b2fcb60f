"""NuggetStep: one pass of the hot path over one batch -- what smoke() and bench.py run.

  non-streaming (w2v_asr.py:528-585):  scorer -> top-k select (+type table) -> row gather ->
      nl x score-biased cross-attention core -> loss -> backward to scorer params, X, S, Q.
  simultaneous  (w2v_simul.py:407-465): scorer(X.detach()) -> quantity loss -> resize -> fire scan ->
      score gather / type table / row gather -> nl x core (ILK optional) -> loss + 0.01*quantity.

Out-of-scope neighbours of the path (wav2vec2 encoder, semantic encoder, decoder self-attention/MLP and
the q_attn/c_attn GEMMs) are replaced by synthetic inputs: X are the encoder states, S the semantic states,
Q_l the per-layer decoder queries, and K_l/V_l are a fixed diagonal linear map of the gathered memory.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import ops
from .nugget import NuggetScorer


@dataclass
class StepInputs:
    x: torch.Tensor                 # [B,T,d] bf16 encoder states (scorer input)
    s: torch.Tensor                 # [B,T,d] bf16 semantic-encoder states (gather source)
    lengths: Sequence[int]          # valid frames per utterance (host)
    lengths_dev: torch.Tensor       # same, int32 on device
    qs: List[torch.Tensor]          # nl x [B,h,L,hd] bf16 decoder queries
    gs: List[torch.Tensor]          # nl x [B,h,L,hd] bf16 fixed cotangents
    kv_scales: List[tuple]          # nl x (ak[d], av[d]) bf16 diagonal K/V maps
    kv_scales_flat: Optional[torch.Tensor] = None       # [2*nl, d] bf16: ak_0, av_0, ak_1, ...
    target_lengths: Optional[Sequence[int]] = None      # simul only (host)
    target_lengths_dev: Optional[torch.Tensor] = None
    cache: Optional[dict] = None    # per-rate device copies of host-derived tensors (kept out of the timed step)
    valid: Optional["ops.ValidFrames"] = None           # ragged batches: lets the scorer skip all-padding row tiles

    def key_len_dev(self, rate):
        """k_b per utterance as an int32 device tensor (computed once per rate, outside any graph capture)."""
        if self.cache is None:
            self.cache = {}
        if rate not in self.cache:
            kb = ops.nugget_counts_host(self.lengths, rate).astype(np.int32)
            self.cache[rate] = (torch.from_numpy(kb).to(self.x.device), int(kb.max()))
        return self.cache[rate]


def make_inputs(lengths, T, d, h, L, nl, device, seed=1234, target_lengths=None, requires_grad=True) -> StepInputs:
    """Seeded synthetic wav2vec2-shaped inputs: X ~ N(0,1) row-normalised, bf16."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    B = len(lengths)
    hd = d // h

    def states():
        t = torch.randn(B, T, d, generator=g)
        t = (t - t.mean(-1, keepdim=True)) / t.std(-1, keepdim=True)
        return t.to(torch.bfloat16).to(device).requires_grad_(requires_grad)

    x, s = states(), states()
    qs = [torch.randn(B, h, L, hd, generator=g).to(torch.bfloat16).to(device).requires_grad_(requires_grad)
          for _ in range(nl)]
    # cotangents live in [B,L,h,hd] memory order (the order the attention kernel writes O in)
    gs = [(torch.randn(B, h, L, hd, generator=g) / (L * hd) ** 0.5).to(torch.bfloat16).permute(0, 2, 1, 3)
          .contiguous().to(device).permute(0, 2, 1, 3) for _ in range(nl)]
    kv = [((1.0 + 0.25 * torch.randn(d, generator=g)).to(torch.bfloat16).to(device),
           (1.0 + 0.25 * torch.randn(d, generator=g)).to(torch.bfloat16).to(device)) for _ in range(nl)]
    ld = torch.tensor(list(lengths), dtype=torch.int32, device=device)
    tl = None if target_lengths is None else torch.tensor(list(target_lengths), dtype=torch.int32, device=device)
    flat = torch.stack([t for pair in kv for t in pair]).contiguous()
    valid = ops.ValidFrames(lengths, T, device) if (min(lengths) < T and x.is_cuda) else None
    return StepInputs(x, s, list(lengths), ld, qs, gs, kv, flat,
                      None if target_lengths is None else list(target_lengths), tl, None, valid)


def _attend(inp: StepInputs, mem, bias, key_len, ilk_diag, with_loss: bool):
    """nl x score-biased cross-attention on K_l/V_l = diag maps of the gathered memory.  Returns (outputs, loss):
    the scalar loss sum_l <O_l, G_l> is only formed when asked for (tests); its gradient is exactly the cotangent
    G_l, which the step feeds to backward directly."""
    B, K, d = mem.shape
    h = inp.qs[0].shape[1]
    kvs = ops.diag_fanout(mem, inp.kv_scales_flat)
    biases = ops.share_bias(bias, len(inp.qs), h)       # one bias-gradient reduction for all layers
    outs = []
    for l, q in enumerate(inp.qs):
        k = kvs[2 * l].view(B, K, h, d // h).permute(0, 2, 1, 3)
        v = kvs[2 * l + 1].view(B, K, h, d // h).permute(0, 2, 1, 3)
        o, _ = ops.xattn(q, k, v, bias=biases[l], key_len=key_len, ilk_diag=ilk_diag)
        outs.append(o)
    loss = None
    if with_loss:
        with torch.no_grad():
            loss = sum((o.float() * g.float()).sum() for o, g in zip(outs, inp.gs))
    return outs, loss, kvs, biases


def nugget_step_asr(scorer: NuggetScorer, inp: StepInputs, rate: float, backward: bool = True,
                    with_loss: bool = False):
    """Non-streaming NuggetStep.  Host-known lengths size the outputs, so the step has no device->host sync
    (and can be captured in a HIP graph).  Returns a dict of the step's observable results."""
    B, T, d = inp.x.shape
    key_len, K = inp.key_len_dev(rate)
    scores = scorer(inp.x, inp.valid).squeeze(-1).float()
    # rows: frames that can receive a score gradient = the B*K selected slots (pad slots get exact zeros)
    nmask, idx, sg, table, pos, rows = ops.topk_select(scores, lengths=inp.lengths_dev, rate=rate, K=K, want_rows=True)
    scorer.set_active_rows(rows)
    mem = ops.gather_rows(inp.s, idx)
    outs, loss, kvs, biases = _attend(inp, mem, sg, key_len, -1, with_loss)
    if backward:
        torch.autograd.backward(outs, inp.gs)
    return {"scores": scores, "nugget_mask": nmask, "indices": idx, "scores_gather": sg, "type_table": table,
            "positions": pos, "mem": mem, "loss": loss, "outs": outs, "kvs": kvs, "biases": biases}


def nugget_step_simul(scorer: NuggetScorer, inp: StepInputs, use_ilk: bool, backward: bool = True,
                      cap: Optional[int] = None, with_loss: bool = False):
    """Simultaneous NuggetStep (CIF-regularised scorer + fire scan, ILK optional).  The number of fires per
    utterance is at most L_b + 1 after `resize`, so cap = max(L_b) + 2 sizes the outputs without a sync."""
    B, T, d = inp.x.shape
    if cap is None:
        cap = int(max(inp.target_lengths)) + 2
    # every valid frame gets the quantity-loss gradient; pad frames get none (alphas = sigmoid(s) * mask), and with
    # inp.valid the scorer's backward runs over the valid frames only
    scorer.set_active_rows(None)
    scores = scorer(inp.x.detach(), inp.valid).squeeze(-1).float()                       # w2v_simul.py:410-411
    # quantity loss (:419-424), resize (:425), fire scan (:426), score gather (:427), type table (:431): one launch
    alphas, qloss, resized, idx, nmask, counts, mx, nugget_scores, table, pos = ops.cif_segment(
        scores, inp.target_lengths_dev, inp.lengths_dev, cap=cap)
    mem = ops.gather_rows(inp.s, idx)                                                    # :446-447
    outs, loss, kvs, biases = _attend(inp, mem, nugget_scores, counts, 2 if use_ilk else -1, with_loss)
    total = None if loss is None else loss + 0.01 * qloss.detach()                       # :461
    if backward:
        torch.autograd.backward(outs + [qloss], inp.gs + [torch.full_like(qloss, 0.01)])
    return {"scores": scores, "alphas": alphas, "resized": resized, "quantity_loss": qloss, "indices": idx,
            "nugget_mask": nmask, "counts": counts, "max_count": mx, "scores_gather": nugget_scores,
            "type_table": table, "positions": pos, "mem": mem, "loss": total, "outs": outs, "kvs": kvs,
            "biases": biases}


def attention_backward(out, inp: StepInputs):
    """First half of a two-stage backward (multi-GPU step): the nl attention backward kernels and the shared
    bias-gradient reduction only.  Returns (d mem, d scores_gather); the decoder-side gradients are complete after
    this stage, so their all-reduce can start while backward_rest() runs."""
    grads = torch.autograd.grad(out["outs"], [out["mem"], out["scores_gather"]] + list(inp.qs), inp.gs)
    return grads[0], grads[1]


def attention_backward_layer(out, inp: StepInputs, l: int):
    """The attention backward of decoder layer l alone (multi-GPU step with one all-reduce bucket per layer, as DDP
    overlaps its buckets with the backward): returns (dK_l, dV_l, d bias_l partials), the cotangents of the layer's
    inputs this side of the shared nodes (K/V fan-out, shared bias), which backward_rest_layers() runs once."""
    k, v, b = out["kvs"][2 * l], out["kvs"][2 * l + 1], out["biases"][l]
    g = torch.autograd.grad([out["outs"][l]], [k, v, b, inp.qs[l]], [inp.gs[l]])
    return g[0], g[1], g[2]


def backward_rest_layers(out, layer_grads, extra=None, extra_grads=None):
    """After attention_backward_layer() for every layer (any order): fan-out backward, the bias-gradient reduction of all
    layers, row-gather backward, score scatter and the scorer backward.  layer_grads[l] = that layer's triple."""
    nl = len(out["outs"])
    roots, gr = [], []
    for l in range(nl):
        roots += [out["kvs"][2 * l], out["kvs"][2 * l + 1]]
        gr += [layer_grads[l][0], layer_grads[l][1]]
    roots += [out["biases"][l] for l in range(nl)]
    gr += [layer_grads[l][2] for l in range(nl)]
    if extra is not None:
        roots, gr = roots + list(extra), gr + list(extra_grads)
    torch.autograd.backward(roots, gr)


def backward_rest(out, dmem, dsg, extra=None, extra_grads=None):
    """Second half: row-gather backward, score scatter and the scorer backward (plus the quantity loss in the
    simultaneous step, passed as extra / extra_grads)."""
    roots, gr = [out["mem"], out["scores_gather"]], [dmem, dsg]
    if extra is not None:
        roots, gr = roots + list(extra), gr + list(extra_grads)
    torch.autograd.backward(roots, gr)
