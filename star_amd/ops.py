"""Operator layer: torch tensors in, HIP kernels (through the C ABI) out.

Every function here launches kernels from libstar_nugget.so on the current torch stream.  torch is
used for device memory, streams and autograd bookkeeping only.  Reference citations are relative to
the STAR checkout (src/models/...).
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Optional, Sequence

import numpy as np
import torch

from . import _lib
from ._lib import XAttnShape, check, lib, ptr, stream_ptr

DT_BF16, DT_F32 = 0, 1


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise StarDeviceError("star_amd ops run on the GPU only (got a CPU tensor); there is no CPU fallback")


class StarDeviceError(RuntimeError):
    pass


def _mask_or_lengths(attn_mask, lengths):
    """Returns (mask_u8 or None, lengths_i32 or None): exactly one is set."""
    if (attn_mask is None) == (lengths is None):
        raise ValueError("give exactly one of attn_mask / lengths")
    if attn_mask is not None:
        m = attn_mask
        if m.dtype != torch.bool and m.dtype != torch.uint8:
            m = m != 0
        m = m.contiguous()
        return m.view(torch.uint8) if m.dtype == torch.bool else m, None
    return None, lengths.to(torch.int32).contiguous()


# ----------------------------------------------------------------------------- a2 + a3
def nugget_counts_host(n_tok, rate: float) -> np.ndarray:
    """k_b = trunc(fp32(n_tok) * fp32(1/rate) + 0.99) clamped to n_tok (w2v_asr.py:219-221), on the host
    (used only to size the outputs; the kernel recomputes k_b itself)."""
    n = np.asarray(n_tok, dtype=np.int64)
    k = (n.astype(np.float32) * np.float32(1.0 / rate)).astype(np.float32) + np.float32(0.99)
    return np.minimum(k.astype(np.float32).astype(np.int64), n)


class _TopkSelect(torch.autograd.Function):
    """apply_nugget_compression + add_nugget_type_embedding (w2v_asr.py:208-244, 316-323) in one kernel.
    Differentiable only through scores_gather (w2v_asr.py:243)."""

    @staticmethod
    def forward(ctx, scores, mask_u8, lengths_i32, n_nugget_i32, ratio, K, want_table, want_rows):
        B, T = scores.shape
        dev = scores.device
        ctx.set_materialize_grads(False)      # no zero tensors for the integer / mask outputs
        s = scores.detach().to(torch.float32).contiguous()
        idx = torch.empty((B, K), dtype=torch.int64, device=dev)
        nmask = torch.empty((B, K), dtype=torch.uint8, device=dev)
        sg = torch.empty((B, K), dtype=torch.float32, device=dev)
        table = torch.empty((B, T), dtype=torch.int64, device=dev) if want_table else None
        pos = torch.empty((B, K), dtype=torch.int64, device=dev) if want_table else None
        rows = torch.empty((B * K,), dtype=torch.int32, device=dev) if want_rows else None
        check(lib().star_topk_select(ptr(s), ptr(mask_u8), ptr(lengths_i32), ptr(n_nugget_i32), float(ratio),
                                     B, T, K, ptr(idx), ptr(nmask), ptr(sg), ptr(table), ptr(pos), ptr(rows),
                                     stream_ptr()), "star_topk_select")
        ctx.save_for_backward(idx)
        ctx.T = T
        ctx.in_dtype = scores.dtype
        nmask = nmask.view(torch.bool)
        ctx.mark_non_differentiable(idx, nmask)
        outs = [nmask, idx, sg.to(scores.dtype)]
        if want_table:
            ctx.mark_non_differentiable(table, pos)
            outs += [table, pos]
        if want_rows:
            ctx.mark_non_differentiable(rows)
            outs.append(rows)
        return tuple(outs)

    @staticmethod
    def backward(ctx, g_mask, g_idx, g_sg, *rest):
        if g_sg is None:
            return None, None, None, None, None, None, None, None
        (idx,) = ctx.saved_tensors
        B, K = idx.shape
        g = g_sg.to(torch.float32).contiguous()
        ds = torch.empty((B, ctx.T), dtype=torch.float32, device=g.device)
        check(lib().star_gather_rows_bwd(ptr(g), ptr(idx), ptr(ds), B, ctx.T, K, 1, DT_F32, stream_ptr()),
              "star_gather_rows_bwd")
        return ds.to(ctx.in_dtype), None, None, None, None, None, None, None


def topk_select(scores, attn_mask=None, lengths=None, rate: Optional[float] = None, n_nugget=None,
                K: Optional[int] = None, want_table: bool = True, want_rows: bool = False):
    """Top-k anchor selection.  `K` (= max_b k_b) may be passed to avoid a device->host sync; when it is
    None it is derived from the lengths exactly like the reference (w2v_asr.py:227: `n_nugget.max()`),
    which costs one small readback.  want_rows appends the flat int32 frame ids b*T + idx (the active-row list of the
    scorer backward) to the returned tuple."""
    _need_cuda(scores, attn_mask, lengths, n_nugget)
    mask_u8, lengths_i32 = _mask_or_lengths(attn_mask, lengths)
    nn_i32 = None if n_nugget is None else n_nugget.to(torch.int32).contiguous()
    if nn_i32 is None and rate is None:
        raise ValueError("need rate or n_nugget")
    ratio = 0.0 if rate is None else float(np.float32(1.0 / rate))
    if K is None:
        if nn_i32 is not None:
            K = int(nn_i32.max().item())
        else:
            n_tok = mask_u8.sum(dim=1) if mask_u8 is not None else lengths_i32
            K = int(nugget_counts_host(n_tok.cpu().numpy(), rate).max())
    return _TopkSelect.apply(scores, mask_u8, lengths_i32, nn_i32, ratio, int(K), want_table, bool(want_rows))


def type_table(nugget_indices, nugget_mask, T: int):
    """add_nugget_type_embedding (w2v_asr.py:316-323): -> (table [B,T] int64, positions [B,K] int64)."""
    _need_cuda(nugget_indices, nugget_mask)
    idx = nugget_indices.to(torch.int64).contiguous()
    m = nugget_mask.contiguous()
    m = m.view(torch.uint8) if m.dtype == torch.bool else (m != 0).view(torch.uint8)
    B, K = idx.shape
    table = torch.empty((B, T), dtype=torch.int64, device=idx.device)
    pos = torch.empty((B, K), dtype=torch.int64, device=idx.device)
    check(lib().star_nugget_type_table(ptr(idx), ptr(m), B, T, K, ptr(table), ptr(pos), stream_ptr()),
          "star_nugget_type_table")
    return table, pos


# ----------------------------------------------------------------------------- a4
class _GatherRows(torch.autograd.Function):
    """semantic_repr[arange(B)[:,None], nugget_indices] (w2v_asr.py:566-567, w2v_simul.py:427,446-447)."""

    @staticmethod
    def forward(ctx, src, idx):
        ctx.set_materialize_grads(False)
        squeeze = src.dim() == 2
        s3 = src.unsqueeze(-1) if squeeze else src
        s3 = s3.contiguous()
        B, T, d = s3.shape
        K = idx.shape[1]
        idx = idx.contiguous()
        out = torch.empty((B, K, d), dtype=s3.dtype, device=s3.device)
        check(lib().star_gather_rows_fwd(ptr(s3), ptr(idx), ptr(out), B, T, K, d * s3.element_size(), stream_ptr()),
              "star_gather_rows_fwd")
        ctx.save_for_backward(idx)
        ctx.shape = (B, T, d)
        ctx.squeeze = squeeze
        return out.squeeze(-1) if squeeze else out

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return None, None
        (idx,) = ctx.saved_tensors
        B, T, d = ctx.shape
        K = idx.shape[1]
        if g.dtype not in (torch.bfloat16, torch.float32):
            raise TypeError(f"gather_rows backward supports bf16/fp32, got {g.dtype}")
        g3 = (g.unsqueeze(-1) if ctx.squeeze else g).contiguous()
        ds = torch.empty((B, T, d), dtype=g.dtype, device=g.device)
        check(lib().star_gather_rows_bwd(ptr(g3), ptr(idx), ptr(ds), B, T, K, d,
                                         DT_BF16 if g.dtype == torch.bfloat16 else DT_F32, stream_ptr()),
              "star_gather_rows_bwd")
        return (ds.squeeze(-1) if ctx.squeeze else ds), None


def gather_rows(src, idx):
    """dst[b,j] = src[b, idx[b,j]] for src [B,T,d] or [B,T]; idx -1 addresses the last frame."""
    _need_cuda(src, idx)
    if src.dtype not in (torch.bfloat16, torch.float32):
        raise TypeError(f"gather_rows: unsupported dtype {src.dtype} (bf16 / fp32, forward and backward)")
    return _GatherRows.apply(src, idx.to(torch.int64))


class _GatherLayerNorm(torch.autograd.Function):
    """ln_f(hidden)[arange(B)[:,None], idx] computed as ln_f(hidden[idx]) (my_transformer.py:945 + w2v_asr.py:566-567):
    only the B*K gathered rows are normalised."""

    @staticmethod
    def forward(ctx, src, idx, weight, bias, eps):
        ctx.set_materialize_grads(False)
        s3 = src.detach().contiguous()
        B, T, d = s3.shape
        K = idx.shape[1]
        idx = idx.contiguous()
        gm = weight.detach().to(torch.float32).contiguous()
        bt = bias.detach().to(torch.float32).contiguous()
        out = torch.empty((B, K, d), dtype=s3.dtype, device=s3.device)
        mean = torch.empty((B, K), dtype=torch.float32, device=s3.device)
        rstd = torch.empty((B, K), dtype=torch.float32, device=s3.device)
        dt = DT_BF16 if s3.dtype == torch.bfloat16 else DT_F32
        check(lib().star_gather_layernorm_fwd(ptr(s3), dt, ptr(idx), ptr(gm), ptr(bt), float(eps), B, T, K, d, ptr(out),
                                              ptr(mean), ptr(rstd), stream_ptr()), "star_gather_layernorm_fwd")
        ctx.save_for_backward(s3, idx, gm, mean, rstd)
        ctx.dt = dt
        ctx.wdtypes = (weight.dtype, bias.dtype)
        return out

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return None, None, None, None, None
        s3, idx, gm, mean, rstd = ctx.saved_tensors
        B, T, d = s3.shape
        K = idx.shape[1]
        g3 = g.to(s3.dtype).contiguous()
        drows = torch.empty((B, K, d), dtype=s3.dtype, device=s3.device)
        dgamma = torch.empty((d,), dtype=torch.float32, device=s3.device)
        dbeta = torch.empty((d,), dtype=torch.float32, device=s3.device)
        wsb = lib().star_gather_layernorm_bwd_workspace(B, K, d)
        ws = torch.empty((max(wsb, 1),), dtype=torch.uint8, device=s3.device)
        check(lib().star_gather_layernorm_bwd(ptr(s3), ctx.dt, ptr(idx), ptr(gm), ptr(mean), ptr(rstd), ptr(g3), B, T, K, d,
                                              ptr(drows), ptr(dgamma), ptr(dbeta), ptr(ws), wsb, stream_ptr()),
              "star_gather_layernorm_bwd")
        dsrc = None
        if ctx.needs_input_grad[0]:
            dsrc = torch.empty((B, T, d), dtype=s3.dtype, device=s3.device)
            check(lib().star_gather_rows_bwd(ptr(drows), ptr(idx), ptr(dsrc), B, T, K, d, ctx.dt, stream_ptr()),
                  "star_gather_rows_bwd")
        return dsrc, None, dgamma.to(ctx.wdtypes[0]), dbeta.to(ctx.wdtypes[1]), None


def gather_layernorm(src, idx, weight, bias, eps: float = 1e-5):
    """LayerNorm(src)[arange(B)[:,None], idx] for src [B,T,d] (bf16 / fp32) without normalising the frames that are not
    gathered: the semantic encoder's ln_f (my_transformer.py:945) followed by the nugget row gather
    (w2v_asr.py:566-567, w2v_simul.py:446-447).  idx -1 addresses the last frame.  Differentiable in src, weight, bias."""
    _need_cuda(src, idx, weight, bias)
    if src.dim() != 3 or src.dtype not in (torch.bfloat16, torch.float32):
        raise TypeError(f"gather_layernorm: src must be [B,T,d] bf16 / fp32, got {tuple(src.shape)} {src.dtype}")
    return _GatherLayerNorm.apply(src, idx.to(torch.int64), weight, bias, float(eps))


class _DiagFanout(torch.autograd.Function):
    """Harness stand-in for the per-layer c_attn projection: K_l/V_l = mem * scale_j (one pass fwd, one bwd)."""

    @staticmethod
    def forward(ctx, mem, scales):
        B, K, d = mem.shape
        J = scales.shape[0]
        m = mem.detach().contiguous()
        sc = scales.detach().to(torch.bfloat16).contiguous()
        out = torch.empty((J, B, K, d), dtype=torch.bfloat16, device=mem.device)
        check(lib().star_diag_fanout_fwd(ptr(m), ptr(sc), J, B * K, d, ptr(out), stream_ptr()), "star_diag_fanout_fwd")
        ctx.save_for_backward(sc)
        ctx.shape = (J, B, K, d)
        return tuple(out[j] for j in range(J))

    @staticmethod
    def backward(ctx, *grads):
        (sc,) = ctx.saved_tensors
        J, B, K, d = ctx.shape
        gs = []
        for g in grads:
            if g is None:
                g = torch.zeros((B, K, d), dtype=torch.bfloat16, device=sc.device)
            gs.append(g.to(torch.bfloat16).contiguous())
        arr = (_lib.C.c_void_p * J)(*[g.data_ptr() for g in gs])
        dmem = torch.empty((B, K, d), dtype=torch.bfloat16, device=sc.device)
        check(lib().star_diag_fanout_bwd(arr, J, ptr(sc), B * K, d, ptr(dmem), stream_ptr()), "star_diag_fanout_bwd")
        return dmem, None


def diag_fanout(mem, scales):
    """mem [B,K,d] bf16, scales [J,d] -> J tensors [B,K,d] (views of one buffer)."""
    _need_cuda(mem, scales)
    if mem.dtype != torch.bfloat16:
        raise TypeError("diag_fanout: bf16 only")
    return _DiagFanout.apply(mem, scales)


# ----------------------------------------------------------------------------- a5
class _CifQuantity(torch.autograd.Function):
    """alphas = sigmoid(scores)*mask; loss = mean((sum alphas/thr - L)^2)  (w2v_simul.py:419-424)."""

    @staticmethod
    def forward(ctx, scores, mask_u8, lengths_i32, target_i32, threshold):
        ctx.set_materialize_grads(False)
        B, T = scores.shape
        s = scores.detach().to(torch.float32).contiguous()
        alphas = torch.empty_like(s)
        diff = torch.empty((B,), dtype=torch.float32, device=s.device)
        loss = torch.empty((1,), dtype=torch.float32, device=s.device)
        check(lib().star_cif_quantity_fwd(ptr(s), ptr(mask_u8), ptr(lengths_i32), ptr(target_i32), float(threshold),
                                          B, T, ptr(alphas), ptr(diff), ptr(loss), stream_ptr()),
              "star_cif_quantity_fwd")
        ctx.save_for_backward(alphas, diff)
        ctx.threshold = float(threshold)
        ctx.in_dtype = scores.dtype
        ctx.mark_non_differentiable(alphas)
        return alphas, loss.reshape(())

    @staticmethod
    def backward(ctx, g_alphas, g_loss):
        if g_loss is None:
            return None, None, None, None, None
        alphas, diff = ctx.saved_tensors
        B, T = alphas.shape
        gl = g_loss.to(torch.float32).reshape(1).contiguous()
        ds = torch.empty_like(alphas)
        check(lib().star_cif_quantity_bwd(ptr(alphas), ptr(diff), ptr(gl), ctx.threshold, B, T, 0, ptr(ds),
                                          stream_ptr()), "star_cif_quantity_bwd")
        return ds.to(ctx.in_dtype), None, None, None, None


def cif_quantity(scores, target_lengths, attn_mask=None, lengths=None, threshold: float = 1.0):
    """Returns (alphas [B,T] fp32 (detached, as the live reference path only uses it detached), loss scalar)."""
    _need_cuda(scores, target_lengths, attn_mask, lengths)
    mask_u8, lengths_i32 = _mask_or_lengths(attn_mask, lengths)
    return _CifQuantity.apply(scores, mask_u8, lengths_i32, target_lengths.to(torch.int32).contiguous(), threshold)


# ----------------------------------------------------------------------------- a6
class _CifResize(torch.autograd.Function):
    """W2VSimulModel.resize (w2v_simul.py:504-535), noise = 0, with its gradient (needed by lm_cif_training,
    w2v_simul.py:558; the nugget path detaches the output, w2v_simul.py:426)."""

    @staticmethod
    def forward(ctx, alphas, target_i32, threshold, max_rounds):
        ctx.set_materialize_grads(False)
        a = alphas.detach().to(torch.float32).contiguous()
        B, T = a.shape
        out = torch.empty_like(a)
        fix = torch.empty((B,), dtype=torch.int32, device=a.device)
        check(lib().star_cif_resize(ptr(a), ptr(target_i32), float(threshold), int(max_rounds), B, T, ptr(out),
                                    ptr(fix), stream_ptr()), "star_cif_resize")
        ctx.save_for_backward(a, target_i32, fix)
        ctx.in_dtype = alphas.dtype
        return out

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return None, None, None, None
        a, tl, fix = ctx.saved_tensors
        B, T = a.shape
        gd = g.to(torch.float32).contiguous()
        da = torch.empty_like(a)
        check(lib().star_cif_resize_bwd(ptr(a), ptr(tl), ptr(fix), ptr(gd), B, T, ptr(da), stream_ptr()),
              "star_cif_resize_bwd")
        return da.to(ctx.in_dtype), None, None, None


def cif_resize(alphas, target_lengths, threshold: float = 1.0, max_rounds: int = 10):
    """W2VSimulModel.resize (w2v_simul.py:504-535), noise = 0.  Differentiable w.r.t. alphas."""
    _need_cuda(alphas, target_lengths)
    return _CifResize.apply(alphas, target_lengths.to(torch.int32).contiguous(), float(threshold), int(max_rounds))


# ----------------------------------------------------------------------------- a7
def cif_fire_scan(alphas, attn_mask=None, lengths=None, cap: Optional[int] = None, scores=None):
    """apply_threshold_yield_nugget's scan (w2v_simul.py:376-403).  Returns
    (idx [B,cap] int64 -1 padded, mask [B,cap] bool, counts [B] int32, max_count [1] int32, scores_gather or None).
    No host sync happens here; `cap` defaults to T."""
    _need_cuda(alphas, attn_mask, lengths, scores)
    mask_u8, lengths_i32 = _mask_or_lengths(attn_mask, lengths)
    a = alphas.detach().to(torch.float32).contiguous()
    B, T = a.shape
    cap = T if cap is None else int(cap)
    dev = a.device
    idx = torch.empty((B, cap), dtype=torch.int64, device=dev)
    m = torch.empty((B, cap), dtype=torch.uint8, device=dev)
    counts = torch.empty((B,), dtype=torch.int32, device=dev)
    mx = torch.empty((1,), dtype=torch.int32, device=dev)
    sc = sg = None
    if scores is not None:
        sc = scores.detach().to(torch.float32).contiguous()
        sg = torch.empty((B, cap), dtype=torch.float32, device=dev)
    check(lib().star_cif_fire_scan(ptr(a), ptr(mask_u8), ptr(lengths_i32), ptr(sc), B, T, cap, ptr(idx), ptr(m),
                                   ptr(counts), ptr(mx), ptr(sg), stream_ptr()), "star_cif_fire_scan")
    return idx, m.view(torch.bool), counts, mx, sg


class _CifSegment(torch.autograd.Function):
    """Quantity loss + resize + fire scan + score gather + type table of the simultaneous path in one launch
    (w2v_simul.py:419-431); gradients flow into the scores through the quantity loss and through scores_gather."""

    @staticmethod
    def forward(ctx, scores, lengths_i32, target_i32, cap, threshold, max_rounds):
        ctx.set_materialize_grads(False)
        B, T = scores.shape
        s = scores.detach().to(torch.float32).contiguous()
        dev = s.device
        alphas, resized = torch.empty_like(s), torch.empty_like(s)
        diff = torch.empty((B,), dtype=torch.float32, device=dev)
        loss = torch.empty((1,), dtype=torch.float32, device=dev)
        fix = torch.empty((B,), dtype=torch.int32, device=dev)
        idx = torch.empty((B, cap), dtype=torch.int64, device=dev)
        m = torch.empty((B, cap), dtype=torch.uint8, device=dev)
        counts = torch.empty((B,), dtype=torch.int32, device=dev)
        mx = torch.empty((1,), dtype=torch.int32, device=dev)
        sg = torch.empty((B, cap), dtype=torch.float32, device=dev)
        table = torch.empty((B, T), dtype=torch.int64, device=dev)
        pos = torch.empty((B, cap), dtype=torch.int64, device=dev)
        check(lib().star_cif_segment(ptr(s), ptr(lengths_i32), ptr(target_i32), float(threshold), int(max_rounds), B, T,
                                     int(cap), ptr(alphas), ptr(diff), ptr(loss), ptr(resized), ptr(fix), ptr(idx), ptr(m),
                                     ptr(counts), ptr(mx), ptr(sg), ptr(table), ptr(pos), stream_ptr()),
              "star_cif_segment")
        ctx.save_for_backward(alphas, diff, idx)
        ctx.threshold = float(threshold)
        ctx.in_dtype = scores.dtype
        mb = m.view(torch.bool)
        ctx.mark_non_differentiable(alphas, resized, idx, mb, counts, mx, table, pos)
        return alphas, loss.reshape(()), resized, idx, mb, counts, mx, sg, table, pos

    @staticmethod
    def backward(ctx, g_alphas, g_loss, g_resized, g_idx, g_mask, g_counts, g_mx, g_sg, g_table, g_pos):
        if g_loss is None and g_sg is None:
            return None, None, None, None, None, None
        alphas, diff, idx = ctx.saved_tensors
        B, T = alphas.shape
        ds = torch.empty_like(alphas)
        if g_sg is not None:          # scatter of the gathered-score gradient: writes every element of ds
            gs = g_sg.to(torch.float32).contiguous()
            check(lib().star_gather_rows_bwd(ptr(gs), ptr(idx), ptr(ds), B, T, idx.shape[1], 1, DT_F32, stream_ptr()),
                  "star_gather_rows_bwd")
        if g_loss is not None:
            gl = g_loss.to(torch.float32).reshape(1).contiguous()
            check(lib().star_cif_quantity_bwd(ptr(alphas), ptr(diff), ptr(gl), ctx.threshold, B, T,
                                              1 if g_sg is not None else 0, ptr(ds), stream_ptr()), "star_cif_quantity_bwd")
        return ds.to(ctx.in_dtype), None, None, None, None, None


def cif_segment(scores, target_lengths, lengths, cap: Optional[int] = None, threshold: float = 1.0, max_rounds: int = 10):
    """The simultaneous path's segmenter (w2v_simul.py:419-431) in one launch, for prefix masks given as lengths:
    returns (alphas, quantity_loss, resized, idx [B,cap] int64 -1 padded, nugget_mask [B,cap] bool, counts, max_count,
    nugget_scores = scores[b, idx] (differentiable), type_table [B,T], positions [B,cap]) -- each bit-identical to
    cif_quantity -> cif_resize -> cif_fire_scan -> gather_rows -> type_table.  No host sync; cap defaults to T."""
    _need_cuda(scores, target_lengths, lengths)
    B, T = scores.shape
    cap = T if cap is None else int(cap)
    return _CifSegment.apply(scores, lengths.to(torch.int32).contiguous(), target_lengths.to(torch.int32).contiguous(),
                             cap, float(threshold), int(max_rounds))


class FireState:
    """Per-utterance state of the streaming fire decision: the fp32 integrator and the frames consumed so far."""

    def __init__(self, batch: int, device):
        self.cur = torch.zeros(batch, dtype=torch.float32, device=device)
        self.pos = torch.zeros(batch, dtype=torch.int32, device=device)


def cif_fire_step(alpha_chunk, lengths, state: FireState, cap: Optional[int] = None, scores=None):
    """Streaming form of cif_fire_scan: feeds the next chunk [B,Tc] of weights; the fires of any
    chunking equal the whole-utterance scan bit for bit.  lengths [B] int32: utterance lengths in absolute frames
    (INT32_MAX while the end is unknown).  Returns (idx [B,cap] int64 ABSOLUTE positions, -1 padded, counts [B] int32,
    scores_gather or None); updates `state` in place.  No host sync."""
    _need_cuda(alpha_chunk, lengths, scores)
    a = alpha_chunk.detach().to(torch.float32).contiguous()
    B, Tc = a.shape
    cap = Tc if cap is None else int(cap)
    if state.cur.shape[0] != B or state.cur.device != a.device:
        raise ValueError("cif_fire_step: state belongs to another batch / device")
    ln = lengths.to(torch.int32).contiguous()
    idx = torch.empty((B, cap), dtype=torch.int64, device=a.device)
    counts = torch.empty((B,), dtype=torch.int32, device=a.device)
    sc = sg = None
    if scores is not None:
        sc = scores.detach().to(torch.float32).contiguous()
        sg = torch.empty((B, cap), dtype=torch.float32, device=a.device)
    check(lib().star_cif_fire_step(ptr(a), ptr(ln), ptr(sc), B, Tc, ptr(state.cur), ptr(state.pos), cap, ptr(idx),
                                   ptr(counts), ptr(sg), stream_ptr()), "star_cif_fire_step")
    return idx, counts, sg


class StreamState:
    """Device-resident state of the streaming segmenter for B concurrent utterances (star_cif_stream_push): integrator,
    frames consumed, newest fire, the nugget memory [B,kmax,d] with its logits and positions, fill counts, the overflow
    counter and the newest frame's row (for the late end-of-utterance flush)."""

    def __init__(self, batch: int, dim: int, kmax: int, device, dtype=torch.bfloat16):
        if dtype not in (torch.bfloat16, torch.float32):
            raise TypeError(f"StreamState: unsupported dtype {dtype}")
        self.B, self.d, self.kmax, self.dtype = int(batch), int(dim), int(kmax), dtype
        z = lambda *sh, dt: torch.zeros(*sh, dtype=dt, device=device)
        self.cur = z(self.B, dt=torch.float32)
        self.pos = z(self.B, dt=torch.int32)
        self.last = torch.full((self.B,), -1, dtype=torch.int32, device=device)
        self.mem = z(self.B, self.kmax, self.d, dt=dtype)
        self.mem_scores = z(self.B, self.kmax, dt=torch.float32)
        self.mem_pos = torch.full((self.B, self.kmax), -1, dtype=torch.int64, device=device)
        self.counts = z(self.B, dt=torch.int32)
        self.dropped = z(self.B, dt=torch.int32)
        self.last_row = z(self.B, self.d, dt=dtype)
        self.last_score = z(self.B, dt=torch.float32)

    def reset(self, rows=None):
        sel = slice(None) if rows is None else rows
        for t, v in ((self.cur, 0), (self.pos, 0), (self.last, -1), (self.counts, 0), (self.dropped, 0), (self.mem_pos, -1)):
            t[sel] = v


def cif_stream_push(scores_chunk, s_chunk, lengths, state: StreamState, idx_out=None, cnt_out=None):
    """One chunk through the streaming segmenter in one launch (star_cif_stream_push): scores_chunk [B,Tc] fp32 scorer
    logits, s_chunk [B,Tc,d] semantic states, lengths [B] int32 absolute utterance lengths (INT32_MAX while unknown).
    scores_chunk = None (and s_chunk = None): the end-of-stream flush alone.  Updates `state` in place; no host sync,
    nothing is allocated here (HIP-graph capturable)."""
    _need_cuda(scores_chunk, s_chunk, lengths)
    if lengths.dtype != torch.int32 or not lengths.is_contiguous():
        raise TypeError("cif_stream_push: lengths must be a contiguous int32 tensor")
    if scores_chunk is None:
        Tc, sp, cp = 0, None, None
    else:
        if scores_chunk.dtype != torch.float32 or not scores_chunk.is_contiguous() or not s_chunk.is_contiguous():
            raise TypeError("cif_stream_push: scores_chunk must be contiguous fp32 and s_chunk contiguous")
        if s_chunk.dtype != state.dtype or s_chunk.shape != (state.B, scores_chunk.shape[1], state.d):
            raise ValueError(f"cif_stream_push: chunk {tuple(s_chunk.shape)} {s_chunk.dtype} does not match the state "
                             f"(B={state.B}, d={state.d}, {state.dtype})")
        Tc, sp, cp = scores_chunk.shape[1], ptr(scores_chunk), ptr(s_chunk)
    if (idx_out is None) != (cnt_out is None):
        raise ValueError("cif_stream_push: idx_out and cnt_out come together")
    if idx_out is not None:
        _need_cuda(idx_out, cnt_out)
        if (idx_out.dtype != torch.int64 or idx_out.dim() != 2 or idx_out.shape[0] != state.B or not idx_out.is_contiguous()
                or cnt_out.dtype != torch.int32 or cnt_out.shape != (state.B,) or not cnt_out.is_contiguous()):
            raise TypeError("cif_stream_push: idx_out must be a contiguous int64 [B, cap] tensor and cnt_out int32 [B]")
    cap = 0 if idx_out is None else idx_out.shape[1]
    check(lib().star_cif_stream_push(sp, cp, DT_BF16 if state.dtype == torch.bfloat16 else DT_F32, ptr(lengths), state.B, Tc,
                                     state.d, ptr(state.cur), ptr(state.pos), ptr(state.last), state.kmax, ptr(state.mem),
                                     ptr(state.mem_scores), ptr(state.mem_pos), ptr(state.counts), ptr(state.dropped),
                                     ptr(state.last_row), ptr(state.last_score), cap, ptr(idx_out), ptr(cnt_out),
                                     stream_ptr()), "star_cif_stream_push")


def cif_block_mask(idx, counts, attn_mask=None, lengths=None, T: Optional[int] = None):
    """block_attn_mask [B,T,T] fp32 (w2v_simul.py:382-402), built on demand from the fire positions."""
    _need_cuda(idx, counts, attn_mask, lengths)
    mask_u8, lengths_i32 = _mask_or_lengths(attn_mask, lengths)
    B, cap = idx.shape
    if T is None:
        T = mask_u8.shape[1]
    out = torch.empty((B, T, T), dtype=torch.float32, device=idx.device)
    check(lib().star_cif_block_mask(ptr(idx.contiguous()), ptr(counts), ptr(mask_u8), ptr(lengths_i32), B, T, cap,
                                    ptr(out), stream_ptr()), "star_cif_block_mask")
    return out


# ----------------------------------------------------------------------------- a1
class ActiveRows:
    """Hint shared between the selector and the scorer: the frames (flattened b*T+t, int32) whose dscore can be
    non-zero in the NEXT forward/backward of the scorer.  In the non-streaming path only the <= K selected frames
    per utterance get a gradient (loss = lm_loss only, w2v_asr.py:580-581), so the scorer backward runs on those
    rows alone.  The hint is SINGLE-USE: it is bound to the scores tensor of the forward that precedes set() (or to
    the next forward when none has run yet) and consumed by that forward's backward; any other forward/backward of
    the same scorer (a dense regulariser, the simultaneous path, a later step that forgets to set it) runs dense.
    dscores outside the listed rows are NOT propagated: list every frame that can receive a gradient."""

    def __init__(self):
        self.rows = None
        self._owner = None          # id of the forward (a counter) the rows belong to
        self._last_forward = 0

    def set(self, rows):
        self.rows = rows
        self._owner = self._last_forward if (rows is not None and self._last_forward > 0) else None

    def _begin_forward(self) -> int:
        self._last_forward += 1
        if self.rows is not None and self._owner is None:
            self._owner = self._last_forward        # set() came before the forward
        return self._last_forward

    def _take(self, forward_id: int):
        """Rows for the backward of forward `forward_id`, or None (dense).  Consumes the hint."""
        if self.rows is None or self._owner != forward_id:
            return None
        rows, self.rows, self._owner = self.rows, None, None
        return rows


# ----------------------------------------------------------------------------- 8f-2
class _CifAggregate(torch.autograd.Function):
    """CIF weighted aggregation (lm_cif_training frame loop, w2v_simul.py:558-606 / w2v_asr.py:681-722)."""

    @staticmethod
    def forward(ctx, alphas, s, threshold, cap):
        ctx.set_materialize_grads(False)
        B, T, d = s.shape
        dev = s.device
        a = alphas.detach().to(torch.float32).contiguous()
        sc = s.detach().contiguous()
        wts = torch.empty((B, T, 4), dtype=torch.float32, device=dev)      # {cur, rem, segment/fire bits, 0} per frame
        idx = torch.empty((B, cap), dtype=torch.int64, device=dev)
        mask = torch.empty((B, cap), dtype=torch.uint8, device=dev)
        counts = torch.empty((B,), dtype=torch.int32, device=dev)
        mx = torch.empty((1,), dtype=torch.int32, device=dev)
        check(lib().star_cif_weights(ptr(a), float(threshold), B, T, cap, ptr(wts), ptr(idx), ptr(mask), ptr(counts),
                                     ptr(mx), stream_ptr()), "star_cif_weights")
        out = torch.empty((B, cap, d), dtype=sc.dtype, device=dev)
        dt = DT_BF16 if sc.dtype == torch.bfloat16 else DT_F32
        check(lib().star_cif_aggregate_fwd(ptr(sc), dt, ptr(wts), ptr(idx), ptr(counts), B, T, cap, d, ptr(out),
                                           stream_ptr()), "star_cif_aggregate_fwd")
        ctx.save_for_backward(sc, wts, counts)
        ctx.cap = cap
        ctx.a_dtype = alphas.dtype
        ctx.mark_non_differentiable(idx, mask, counts, mx)
        return out, mask, idx, counts, mx

    @staticmethod
    def backward(ctx, g, *unused):
        if g is None:
            return None, None, None, None
        sc, wts, counts = ctx.saved_tensors
        B, T, d = sc.shape
        gc = g.to(sc.dtype).contiguous()
        ds = torch.empty_like(sc)
        da = torch.empty((B, T), dtype=torch.float32, device=sc.device)
        ws = torch.empty((2 * B * T,), dtype=torch.float32, device=sc.device)
        dt = DT_BF16 if sc.dtype == torch.bfloat16 else DT_F32
        check(lib().star_cif_aggregate_bwd(ptr(gc), ptr(sc), dt, ptr(wts), ptr(counts), B, T, ctx.cap, d, ptr(ds),
                                           ptr(da), ptr(ws), stream_ptr()), "star_cif_aggregate_bwd")
        return da.to(ctx.a_dtype), ds, None, None


def cif_aggregate(alphas, s, threshold: float = 1.0, cap: Optional[int] = None):
    """alphas [B,T] (masked, resized), s [B,T,d] bf16/fp32 -> (cif_repr [B,cap,d], cif_mask [B,cap] uint8,
    fired_positions [B,cap] int64 (-1 padded), counts [B] int32, max_count [1] int32).  Differentiable w.r.t. alphas
    and s.  `cap` (>= the largest fire count) avoids a device->host sync; when None it is T (then trim with
    max_count, as the reference sizes its output with num_fires.max())."""
    _need_cuda(alphas, s)
    if s.dtype not in (torch.bfloat16, torch.float32):
        raise TypeError(f"cif_aggregate: unsupported dtype {s.dtype}")
    if alphas.shape != s.shape[:2]:
        raise ValueError(f"cif_aggregate: alphas {tuple(alphas.shape)} does not match states {tuple(s.shape)}")
    return _CifAggregate.apply(alphas, s, float(threshold), int(cap if cap is not None else s.shape[1]))


_AUX_STREAMS = {}
_AUX_LOCK = __import__("threading").Lock()


def _aux_stream(dev):
    """Second stream + the four fork/join events handed to star_scorer_bwd (one set per device, calling stream and
    calling thread: autograd runs backward on its own thread, and two threads must not share events)."""
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream, __import__("threading").get_ident())
    with _AUX_LOCK:
        ent = _AUX_STREAMS.get(key)
        if ent is None:
            st = torch.cuda.Stream(device=dev)
            evs = [torch.cuda.Event(enable_timing=False) for _ in range(4)]
            for e in evs:
                e.record(st)                      # torch creates the hipEvent_t lazily, at the first record
            arr = (_lib.C.c_void_p * 4)(*[e.cuda_event for e in evs])
            ent = (st, evs, arr)
            _AUX_STREAMS[key] = ent
    return ent


class ValidFrames:
    """Host-known lengths of a padded batch [B,T,...] prepared once for the ragged scorer: the device copy of the
    lengths, the flat ids of the valid frames (the rows the compact forward gathers, and the active rows of a backward
    that reaches every valid frame, as the quantity loss does), their exclusive prefix sums, the frame -> compact-row
    table, and the 256-row tiles of the flattened [B*T] frame axis that hold a valid frame (what the tile form,
    star_scorer_fwd_ragged, computes).  Built from a host list, so using it costs no device->host sync (HIP-graph
    safe).  compact=True (default): the scorer GEMM runs on the valid frames only (star_scorer_fwd_compact);
    compact=False: on the live 256-row tiles of the padded layout."""

    TILE = 256

    def __init__(self, lengths: Sequence[int], T: int, device, compact: bool = True):
        ln = np.minimum(np.asarray(list(lengths), dtype=np.int64), T)
        self.B, self.T = int(ln.size), int(T)
        self.compact = bool(compact)
        starts = np.arange(self.B, dtype=np.int64) * T
        live = set()
        for b in range(self.B):
            if ln[b] > 0:
                live.update(range(int(starts[b] // self.TILE), int((starts[b] + ln[b] - 1) // self.TILE) + 1))
        # (one dummy entry keeps the pointer non-null for a batch without a single valid frame; n_live says 0)
        self.live_tiles = torch.tensor(sorted(live) or [0], dtype=torch.int32, device=device)
        self.n_live = len(live)
        rows = np.concatenate([starts[b] + np.arange(ln[b]) for b in range(self.B)]) if ln.sum() else np.zeros(0)
        self.n_valid = int(ln.sum())
        # (rows / arange keep one dummy entry for an all-padding batch, as live_tiles does)
        self.rows = torch.from_numpy(rows.astype(np.int32)).to(device)
        self._rows_ptr = self.rows if self.n_valid else torch.zeros(1, dtype=torch.int32, device=device)
        self.lengths = torch.from_numpy(ln.astype(np.int32)).to(device)
        offs = np.concatenate([[0], np.cumsum(ln)[:-1]])
        self.offsets = torch.from_numpy(offs.astype(np.int32)).to(device)
        # frame id -> row of the compacted list (pad frames -> row 0: they only ever carry a zero score gradient)
        table = np.zeros(self.B * self.T, dtype=np.int32)
        table[rows.astype(np.int64)] = np.arange(self.n_valid, dtype=np.int32)
        self.row_of_frame = torch.from_numpy(table).to(device)
        self.arange = torch.arange(max(self.n_valid, 1), dtype=torch.int32, device=device)[:self.n_valid]

    def matches(self, x) -> bool:
        return x.dim() == 3 and x.shape[0] == self.B and x.shape[1] == self.T

    def bit_rows(self, rows):
        """compact-list rows of the frames `rows` (int32 frame ids): where the compact forward left their ReLU bits"""
        if rows is self.rows:
            return self.arange
        return self.row_of_frame[rows.long()]


class _Scorer(torch.autograd.Function):
    """scores = w2 . relu(W1 x + b1) + b2 (w2v_asr.py:113-117,553); bf16 MFMA, fp32 accumulate/scores."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, hint, valid=None):
        ctx.set_materialize_grads(False)
        if valid is not None and not valid.matches(x):
            raise ValueError(f"scorer: ValidFrames for [{valid.B},{valid.T}] used with input {tuple(x.shape)}")
        d = x.shape[-1]
        xb = x.detach()
        if xb.dtype != torch.bfloat16:
            xb = xb.to(torch.bfloat16)
        xb = xb.contiguous().view(-1, d)
        N = xb.shape[0]
        dev = xb.device
        w1f = w1.detach().to(torch.float32).contiguous()
        b1f = b1.detach().to(torch.float32).contiguous()
        w2f = w2.detach().to(torch.float32).contiguous().view(-1)
        b2f = b2.detach().to(torch.float32).contiguous().view(-1)
        need_grad = any(ctx.needs_input_grad[:5])     # all False under no_grad / for detached inputs
        need_dx = need_grad and ctx.needs_input_grad[0]
        w1b = torch.empty((d, d), dtype=torch.bfloat16, device=dev)
        # the backward's dX GEMM wants (diag(w2) W1)^T in bf16: written by the same pass as the bf16 copy of W1
        w1ts = torch.empty((d, d), dtype=torch.bfloat16, device=dev) if need_dx else None
        check(lib().star_cast_transpose_bf16(ptr(w1f), d, d, ptr(w1b), ptr(w1ts), ptr(w2f) if need_dx else None,
                                             stream_ptr()), "star_cast_transpose_bf16")
        wsb = lib().star_scorer_fwd_workspace(N, d)
        ws = torch.empty((wsb,), dtype=torch.uint8, device=dev)
        # ReLU sign bits (1 bit per element of H): all the backward needs of the hidden activation
        bits = torch.empty((lib().star_scorer_relu_bits_bytes(N, d),), dtype=torch.uint8, device=dev) if need_grad else None
        scores = torch.empty((N,), dtype=torch.float32, device=dev)
        if valid is None:
            check(lib().star_scorer_fwd(ptr(xb), ptr(w1b), ptr(b1f), ptr(w2f), ptr(b2f), N, d, ptr(scores), ptr(bits),
                                        ptr(ws), wsb, stream_ptr()), "star_scorer_fwd")
        elif valid.compact:
            check(lib().star_scorer_fwd_compact(ptr(xb), ptr(w1b), ptr(b1f), ptr(w2f), ptr(b2f), ptr(valid.lengths),
                                                ptr(valid.offsets), ptr(valid._rows_ptr), valid.n_valid, valid.B, valid.T,
                                                d, ptr(scores), ptr(bits), ptr(ws), wsb, stream_ptr()),
                  "star_scorer_fwd_compact")
        else:
            check(lib().star_scorer_fwd_ragged(ptr(xb), ptr(w1b), ptr(b1f), ptr(w2f), ptr(b2f), ptr(valid.lengths),
                                               valid.B, valid.T, d, ptr(valid.live_tiles), valid.n_live, ptr(scores),
                                               ptr(bits), ptr(ws), wsb, stream_ptr()), "star_scorer_fwd_ragged")
        ctx.valid = valid
        ctx.save_for_backward(xb, w1b, w1ts, b1f, w2f, bits)
        ctx.hint = hint
        ctx.forward_id = hint._begin_forward() if hint is not None else 0
        ctx.x_shape = x.shape
        ctx.x_dtype = x.dtype
        ctx.w_dtypes = (w1.dtype, b1.dtype, w2.dtype, b2.dtype)
        ctx.w2_shape = w2.shape
        ctx.b2_shape = b2.shape
        return scores.view(x.shape[:-1])

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return None, None, None, None, None, None, None
        xb, w1b, w1ts, b1f, w2f, bits = ctx.saved_tensors
        N, d = xb.shape
        dev = xb.device
        ds = g.to(torch.float32).contiguous().view(-1)
        rows = ctx.hint._take(ctx.forward_id) if ctx.hint is not None else None
        if rows is not None and rows.numel() > N:
            raise ValueError(f"scorer backward: {rows.numel()} active rows for {N} frames (stale hint?)")
        if rows is None:
            # ragged forward: only valid frames have ReLU bits (and only they can carry a score gradient that matters:
            # pad scores are constants)
            rows = ctx.valid.rows if ctx.valid is not None else torch.arange(N, dtype=torch.int32, device=dev)
        # the compact forward left the ReLU bits indexed by row of the valid-frame list, not by frame
        bit_rows = ctx.valid.bit_rows(rows) if (ctx.valid is not None and ctx.valid.compact) else None
        rows = rows.to(torch.int32).contiguous().view(-1)
        if bit_rows is not None:
            bit_rows = bit_rows.to(torch.int32).contiguous().view(-1)
        R = rows.numel()
        if R == 0:
            z = lambda t, dt: torch.zeros(t, dtype=dt, device=dev)
            t1, t2, t3, t4 = ctx.w_dtypes
            gx = z(ctx.x_shape, ctx.x_dtype) if ctx.needs_input_grad[0] else None
            return gx, z((d, d), t1), z((d,), t2), z(ctx.w2_shape, t3), z(ctx.b2_shape, t4), None, None
        need_dx = ctx.needs_input_grad[0] and w1ts is not None
        dx = torch.empty((N, d), dtype=torch.bfloat16, device=dev) if need_dx else None   # zero-filled by the call
        dw1 = torch.empty((d, d), dtype=torch.float32, device=dev)
        db1 = torch.empty((d,), dtype=torch.float32, device=dev)
        dw2 = torch.empty((d,), dtype=torch.float32, device=dev)
        db2 = torch.empty((1,), dtype=torch.float32, device=dev)
        wsb = lib().star_scorer_bwd_workspace(R, d)
        ws = torch.empty((wsb,), dtype=torch.uint8, device=dev)
        # the fork/join costs four event operations on the host: worth it when the launches are replayed from a HIP
        # graph (no host cost at replay), not for eager launches, which are host-bound already
        aux = _aux_stream(dev) if torch.cuda.is_current_stream_capturing() else None
        check(lib().star_scorer_bwd_mapped(ptr(xb), ptr(w1b), ptr(w1ts), ptr(b1f), ptr(w2f), ptr(bits), ptr(ds), ptr(rows),
                                           ptr(bit_rows), R, N, d, ptr(dw1), ptr(db1), ptr(dw2), ptr(db2), ptr(dx), 0,
                                           ptr(ws), wsb, stream_ptr(), aux[0].cuda_stream if aux is not None else None,
                                           aux[2] if aux is not None else None), "star_scorer_bwd_mapped")
        t1, t2, t3, t4 = ctx.w_dtypes
        gx = dx.view(ctx.x_shape).to(ctx.x_dtype) if need_dx else None
        return (gx, dw1.to(t1), db1.to(t2), dw2.view(ctx.w2_shape).to(t3), db2.view(ctx.b2_shape).to(t4), None, None)


def scorer(x, w1, b1, w2, b2, hint: Optional[ActiveRows] = None, valid: Optional[ValidFrames] = None):
    """x [..., d] -> scores [...] fp32.  w1 [d,d] (nn.Linear layout), b1 [d], w2 [1,d], b2 [1].
    valid (x [B,T,d] padded): pad frames get score 0 and are not computed: the GEMM runs on the gathered valid frames
    (star_scorer_fwd_compact) or, for ValidFrames(compact=False), on the 256-row tiles that hold a valid frame
    (star_scorer_fwd_ragged); the reference scores them and masks the result everywhere it is used."""
    _need_cuda(x, w1, b1, w2, b2)
    return _Scorer.apply(x, w1, b1, w2, b2, hint, valid)


# ----------------------------------------------------------------------------- a8
class _DropoutState:
    """Per-device {seed, offset} pair for attention dropout, kept ON THE DEVICE so that a step captured in a HIP
    graph draws a fresh mask at every replay: every forward snapshots the pair (the backward regenerates the mask
    from the snapshot) and advances the offset with a stream-ordered add.

    The pair of a device is ONE tensor for the life of the process: a re-seed (torch.manual_seed /
    torch.cuda.manual_seed with a new value, or reset_dropout_state) is written INTO it with copy_, so a graph that was
    captured earlier sees the new seed at its next replay.  The tensor has to exist before a capture starts -- creating
    it is an allocation plus a host-to-device copy, which a capture cannot hold: call init_dropout_state(device) (or
    run one dropout forward) first; next() raises instead of allocating under capture.  A seed that changed cannot be
    picked up DURING capture either (the copy_ would be recorded into the graph and replayed every time): that raises
    too.  Ranks of a torch.distributed job draw different masks: the rank is folded into the seed."""
    _states = {}            # device index -> [seed it was last written with, tensor]
    _forced_seed = None

    @staticmethod
    def _rank():
        import torch.distributed as td
        return td.get_rank() if td.is_available() and td.is_initialized() else 0

    @classmethod
    def _seed_now(cls):
        seed = cls._forced_seed if cls._forced_seed is not None else torch.cuda.initial_seed()
        # (splitmix-style odd multiplier: distinct ranks -> distinct 63-bit seeds, rank 0 keeps the plain seed)
        return (seed + cls._rank() * 0x9E3779B97F4A7C15) & 0x7FFFFFFFFFFFFFFF

    @classmethod
    def state(cls, dev, restart=False):
        seed = cls._seed_now()
        ent = cls._states.get(dev.index)
        capturing = torch.cuda.is_current_stream_capturing()
        if ent is None:
            if capturing:
                raise _lib.StarNuggetError(
                    "attention dropout: the device-side {seed, offset} state of this GPU does not exist yet and cannot "
                    "be created during HIP-graph capture; call star_amd.ops.init_dropout_state(device) before capturing")
            ent = [seed, torch.tensor([seed, 0], dtype=torch.int64, device=dev)]
            cls._states[dev.index] = ent
        elif ent[0] != seed or restart:
            if capturing:
                raise _lib.StarNuggetError(
                    "attention dropout: the seed changed (torch.manual_seed / reset_dropout_state) while a HIP graph is "
                    "being captured; re-seed before the capture (replays of an existing graph pick a new seed up)")
            ent[1].copy_(torch.tensor([seed, 0], dtype=torch.int64), non_blocking=False)
            ent[0] = seed
        return ent[1]

    @classmethod
    def next(cls, dev):
        st = cls.state(dev)
        snap = st.clone()
        st[1:].add_(1)
        return snap


def init_dropout_state(device=None):
    """Create (or re-seed) the attention-dropout state of `device` now -- required once before a step that uses dropout
    is captured in a HIP graph, harmless otherwise."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    _DropoutState.state(dev)


def reset_dropout_state(seed: Optional[int] = None):
    """Restart the attention-dropout stream of every device that has one: from `seed`, or (None) from torch's CUDA seed
    (torch.manual_seed / torch.cuda.manual_seed with a NEW value re-seed it on their own).  The device tensors are
    rewritten in place, so graphs captured before the call replay with the new stream."""
    _DropoutState._forced_seed = None if seed is None else int(seed)
    for idx in list(_DropoutState._states):
        _DropoutState.state(torch.device("cuda", idx), restart=True)


def _xattn_shape(q, k, v, o, bias_len, ilk_diag, scale, dropout=None):
    B, h, L, hd = q.shape
    Kn = k.shape[2]
    for t, name in ((q, "q"), (k, "k"), (v, "v"), (o, "o")):
        if t.stride(-1) != 1:
            raise ValueError(f"xattn: {name} must have a contiguous head dimension")
        if t.data_ptr() % 16:
            raise ValueError(f"xattn: {name} must be 16-byte aligned")
    s = XAttnShape()
    s.B, s.h, s.L, s.Kn, s.hd = B, h, L, Kn, hd
    s.q_sb, s.q_sh, s.q_sr = q.stride(0), q.stride(1), q.stride(2)
    s.k_sb, s.k_sh, s.k_sr = k.stride(0), k.stride(1), k.stride(2)
    s.v_sb, s.v_sh, s.v_sr = v.stride(0), v.stride(1), v.stride(2)
    s.o_sb, s.o_sh, s.o_sr = o.stride(0), o.stride(1), o.stride(2)
    s.bias_len, s.ilk_diag, s.scale = int(bias_len), int(ilk_diag), float(scale)
    s.dropout_p, s.seed, s.offset, s.rng_state = 0.0, 0, 0, None
    if dropout is not None:
        p, rng = dropout
        s.dropout_p = float(p)
        if isinstance(rng, torch.Tensor):
            s.rng_state = rng.data_ptr()
        else:
            s.seed, s.offset = int(rng[0]) & 0xFFFFFFFFFFFFFFFF, int(rng[1]) & 0xFFFFFFFFFFFFFFFF
    return s


def _empty_same_strides(t):
    e = torch.empty_like(t)
    if e.stride() == t.stride():
        return e
    if any(st == 0 and n > 1 for st, n in zip(t.stride(), t.shape)):
        raise _lib.StarNuggetError("xattn: expanded (stride-0) q / k / v are not supported; call .contiguous() first")
    return torch.empty_strided(t.shape, t.stride(), dtype=t.dtype, device=t.device)


class _XAttn(torch.autograd.Function):
    """softmax(QK^T*scale + ILK + bias) V with d(bias) = sum_{h,i} dS (my_transformer.py:535-603)."""

    @staticmethod
    def forward(ctx, q, k, v, bias, key_len, ilk_diag, scale, need_probs, dropout_p, dropout_rng, kv_grad=None):
        ctx.set_materialize_grads(False)
        in_dtype = q.dtype
        # a slot of share_kv()'s gradient buffer: usable only when the kernels see k / v themselves (no dtype copy)
        ctx.kv_grad = kv_grad if (k.dtype == torch.bfloat16 and v.dtype == torch.bfloat16) else None
        qb, kb, vb = (t.detach() if t.dtype == torch.bfloat16 else t.detach().to(torch.bfloat16) for t in (q, k, v))
        B, h, L, hd = qb.shape
        dev = qb.device
        # output in [B, L, h, hd] memory order (what merge_heads wants), exposed as [B,h,L,hd]
        o = torch.empty((B, L, h, hd), dtype=torch.bfloat16, device=dev).permute(0, 2, 1, 3)
        lse = torch.empty((B, h, L), dtype=torch.float32, device=dev)
        per_head = bias is not None and bias.dim() == 3
        if per_head:
            # [B,h,K] view made by share_bias(): the same [B,K] logits for every head (stride 0 over heads); the
            # gradient then leaves as the per-head partials and share_bias() reduces all layers in one launch
            if bias.shape[1] != h or (h > 1 and bias.stride(1) != 0):
                raise ValueError("xattn: a 3-d bias must be a [B,h,K] broadcast of [B,K] logits (see share_bias)")
            bias = bias[:, 0, :]
        bias_f = None if bias is None else bias.detach().to(torch.float32).contiguous()
        bias_len = 0 if bias_f is None else bias_f.shape[1]
        ctx.bias_per_head = per_head
        kl = None if key_len is None else key_len.to(torch.int32).contiguous()
        probs = torch.empty((B, h, L, kb.shape[2]), dtype=torch.float32, device=dev) if need_probs else None
        dropout = None
        if dropout_p > 0.0:
            # (seed, offset) host integers (reproducible tests) or the per-device state advanced on the stream
            rng = dropout_rng if dropout_rng is not None else _DropoutState.next(dev)
            dropout = (float(dropout_p), rng)
        s = _xattn_shape(qb, kb, vb, o, bias_len, ilk_diag, scale, dropout)
        check(lib().star_xattn_fwd(ptr(qb), ptr(kb), ptr(vb), ptr(bias_f), ptr(kl), s, ptr(o), ptr(lse), ptr(probs),
                                   stream_ptr()), "star_xattn_fwd")
        ctx.save_for_backward(qb, kb, vb, o, lse, bias_f, kl)
        ctx.cfg = (ilk_diag, scale, bias_len)
        ctx.dropout = dropout
        ctx.in_dtype = in_dtype
        ctx.bias_dtype = None if bias is None else bias.dtype
        out = o if in_dtype == torch.bfloat16 else o.to(in_dtype)
        if need_probs:
            ctx.mark_non_differentiable(probs)
            return out, probs
        return out, None

    @staticmethod
    def backward(ctx, g, g_probs):
        if g is None:
            return (None,) * 11
        qb, kb, vb, o, lse, bias_f, kl = ctx.saved_tensors
        ilk_diag, scale, bias_len = ctx.cfg
        B, h, L, hd = qb.shape
        Kn = kb.shape[2]
        dev = qb.device
        # dO must share o's strides
        if g.dtype == torch.bfloat16 and g.stride() == o.stride():
            go = g
        else:
            go = torch.empty_like(o)
            go.copy_(g)
        # the kernels write dQ / dK / dV with the strides of Q / K / V.  Views with gaps (K and V are the two halves of
        # one c_attn output in MyGPT2Attention.forward, my_transformer.py:500-508) are not reproduced by empty_like.
        dq = _empty_same_strides(qb)
        dk = dv = None
        if ctx.kv_grad is not None:
            # K / V are views of one projection output shared by all decoder layers: dK / dV go straight into the
            # matching views of the shared gradient buffer (no per-layer allocation of the whole span, no copies)
            dk, dv = ctx.kv_grad.claim(kb, vb)
        if dk is None:
            dk, dv = _empty_same_strides(kb), _empty_same_strides(vb)
        delta = torch.empty((B, h, L), dtype=torch.float32, device=dev)
        dbias = part = None
        if bias_f is not None:
            part = torch.empty((B, h, Kn), dtype=torch.float32, device=dev)
            if not ctx.bias_per_head:
                dbias = torch.empty((B, bias_len), dtype=torch.float32, device=dev)
        s = _xattn_shape(qb, kb, vb, o, bias_len, ilk_diag, scale, ctx.dropout)
        check(lib().star_xattn_bwd(ptr(qb), ptr(kb), ptr(vb), ptr(o), ptr(go), ptr(lse), ptr(bias_f), ptr(kl), s,
                                   ptr(dq), ptr(dk), ptr(dv), ptr(delta), ptr(part), ptr(dbias), 0, stream_ptr()),
              "star_xattn_bwd")
        dt = ctx.in_dtype
        if dt != torch.bfloat16:
            dq, dk, dv = dq.to(dt), dk.to(dt), dv.to(dt)
        if ctx.bias_per_head:
            dbias = part if bias_len == Kn else part[:, :, :bias_len]
        if dbias is not None and ctx.bias_dtype != torch.float32:
            dbias = dbias.to(ctx.bias_dtype)
        return dq, dk, dv, dbias, None, None, None, None, None, None, None


class _TypeEmbedAdd(torch.autograd.Function):
    """x + type_embed(table) in one pass (w2v_asr.py:560); d x = d out, d weight by a two-stage column reduction."""

    @staticmethod
    def forward(ctx, x, table, weight):
        ctx.set_materialize_grads(False)
        xd, wd = x.detach().contiguous(), weight.detach().to(x.dtype).contiguous()
        tb = table.detach().to(torch.int64).contiguous()
        d = xd.shape[-1]
        out = torch.empty_like(xd)
        dt = DT_BF16 if xd.dtype == torch.bfloat16 else DT_F32
        check(lib().star_type_embed_add(ptr(xd), ptr(tb), ptr(wd), dt, xd.numel() // d, d, ptr(out), stream_ptr()),
              "star_type_embed_add")
        ctx.save_for_backward(tb)
        ctx.meta = (dt, d, weight.dtype)
        return out

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return None, None, None
        (tb,) = ctx.saved_tensors
        dt, d, wdtype = ctx.meta
        dw = None
        if ctx.needs_input_grad[2]:
            gd = g.detach()
            want = torch.bfloat16 if dt == DT_BF16 else torch.float32
            if gd.dtype != want or not gd.is_contiguous():
                gd = gd.to(want).contiguous()
            n = gd.numel() // d
            nbytes = lib().star_type_embed_grad_workspace(n, d, dt)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=gd.device)
            dwf = torch.empty((2, d), dtype=torch.float32, device=gd.device)
            check(lib().star_type_embed_grad(ptr(gd), ptr(tb), dt, n, d, ptr(dwf), ptr(ws), nbytes, stream_ptr()),
                  "star_type_embed_grad")
            dw = dwf.to(wdtype)
        return (g if ctx.needs_input_grad[0] else None), None, dw


def type_embed_add(x, table, weight):
    """inputs_embed = x + weight[table]  (x [..., d] bf16/fp32, table [...] 0/1, weight [2, d])."""
    _need_cuda(x, table, weight)
    if x.dtype not in (torch.bfloat16, torch.float32):
        raise TypeError("type_embed_add: bf16 or fp32 only")
    if weight.dim() != 2 or weight.shape[0] != 2 or weight.shape[1] != x.shape[-1]:
        raise ValueError(f"type_embed_add: weight must be [2, {x.shape[-1]}], got {tuple(weight.shape)}")
    if tuple(table.shape) != tuple(x.shape[:-1]):
        raise ValueError(f"type_embed_add: table {tuple(table.shape)} does not match x {tuple(x.shape)}")
    if x.shape[-1] % (8 if x.dtype == torch.bfloat16 else 4):
        raise ValueError("type_embed_add: the hidden size must be a multiple of 16 bytes")
    return _TypeEmbedAdd.apply(x, table, weight)


class _ShareBias(torch.autograd.Function):
    """One [B,K] tensor of scorer logits handed to nl decoder layers (my_transformer.py:917-928): forward gives every
    layer a [B,h,K] broadcast view, backward sums the layers' per-head gradient partials in ONE launch
    (star_xattn_dbias_sum) instead of nl head reductions plus nl-1 additions by autograd."""

    @staticmethod
    def forward(ctx, bias, nl, h):
        ctx.set_materialize_grads(False)
        b = bias.detach()
        ctx.meta = (nl, h, tuple(bias.shape), bias.dtype)
        return tuple(b.unsqueeze(1).expand(b.shape[0], h, b.shape[1]) for _ in range(nl))

    @staticmethod
    def backward(ctx, *grads):
        nl, h, (B, K), dtype = ctx.meta
        parts = [g for g in grads if g is not None]
        if not parts:
            return None, None, None
        fixed = []
        for g in parts:
            # the attention backward hands over its [B,h,Kn] scratch (possibly a [:, :, :K] slice of it)
            if g.dtype != torch.float32 or g.stride(2) != 1 or g.stride(0) != h * g.stride(1):
                g = g.to(torch.float32).contiguous()
            fixed.append(g)
        kn = {g.stride(1) for g in fixed}
        if len(kn) != 1 or len(fixed) > 32:
            out = sum(g.sum(1) for g in fixed)                      # mixed layouts: plain torch reduction
        else:
            out = torch.empty((B, K), dtype=torch.float32, device=fixed[0].device)
            arr = (_lib.C.c_void_p * len(fixed))(*[g.data_ptr() for g in fixed])
            check(lib().star_xattn_dbias_sum(arr, len(fixed), B, h, kn.pop(), K, ptr(out), stream_ptr()),
                  "star_xattn_dbias_sum")
        return out.to(dtype), None, None


def share_bias(bias, n_layers: int, n_heads: int):
    """bias [B,K] -> tuple of n_layers [B,h,K] broadcast views to pass as `bias` of xattn(), one per decoder layer."""
    _need_cuda(bias)
    if bias.dim() != 2:
        raise ValueError("share_bias: bias must be [B, K]")
    return _ShareBias.apply(bias, int(n_layers), int(n_heads))


class _KVGradBuffer:
    """dK / dV of nl attention layers in ONE [B, Kn, nl*2*d] buffer laid out like the shared projection output, so
    that the projection's backward is two GEMMs over all layers.  Allocated when the first gradient arrives, released
    by the projection's backward."""

    def __init__(self, shape, dtype, device, nl, h):
        self.shape, self.dtype, self.device, self.nl, self.h = tuple(shape), dtype, device, nl, h
        self.buf = None
        self.claimed = set()
        self.stats = {"in_place": 0, "copied": 0, "zeroed": 0}      # how the last backward's gradients arrived (tests)

    def flat(self):
        if self.buf is None:
            self.buf = torch.empty(self.shape, dtype=self.dtype, device=self.device)
        return self.buf

    def view(self, i):
        """view i = 2 * layer + (0 key | 1 value) as [B, h, Kn, hd]"""
        return _kv_views(self.flat(), self.nl, self.h)[i]

    def release(self):
        self.buf = None
        self.claimed.clear()


class KVGradSlot:
    """Where one layer's dK / dV belong in the shared gradient buffer (handed to xattn(kv_grad=...))."""

    def __init__(self, buffer: _KVGradBuffer, layer: int):
        self.buffer, self.layer = buffer, layer

    def claim(self, kb, vb):
        """-> (dk, dv) views of the shared buffer with the strides of kb / vb, or (None, None) when the slot was
        already handed out in this backward (the same K / V used by two attention calls: autograd has to add the
        two gradients, so the second one needs memory of its own) or does not match."""
        b = self.buffer
        i = 2 * self.layer
        if i in b.claimed or kb.dtype != b.dtype:
            return None, None
        dk, dv = b.view(i), b.view(i + 1)
        if dk.shape != kb.shape or dk.stride() != kb.stride() or dv.shape != vb.shape or dv.stride() != vb.stride():
            return None, None
        b.claimed.add(i)
        return dk, dv


def _kv_views(kv, nl, h):
    """kv [B, Kn, nl*2*d] -> [k_0, v_0, k_1, v_1, ...] as [B, h, Kn, hd] views (split_heads of the two halves of
    every layer's c_attn output, my_transformer.py:500-508)."""
    B, Kn, n = kv.shape
    hd = n // (2 * nl * h)
    t = kv.view(B, Kn, nl, 2, h, hd)
    return [t[:, :, l, i].permute(0, 2, 1, 3) for l in range(nl) for i in range(2)]


class _SplitKV(torch.autograd.Function):
    """split + split_heads of a c_attn output that holds the K | V columns of nl layers side by side
    (my_transformer.py:500-508): forward hands out 2*nl [B,h,Kn,hd] views; backward returns ONE [B,Kn,nl*2*d]
    gradient that the attention backwards assembled in place (KVGradSlot), so the projection's backward is two GEMMs
    over all layers and nothing is concatenated."""

    @staticmethod
    def forward(ctx, kv, nl, h, gbuf):
        ctx.set_materialize_grads(False)
        ctx.gbuf = gbuf
        return tuple(_kv_views(kv.detach(), nl, h))

    @staticmethod
    def backward(ctx, *grads):
        gbuf = ctx.gbuf
        if all(g is None for g in grads):
            gbuf.release()
            return None, None, None, None
        gbuf.stats = {"in_place": 0, "copied": 0, "zeroed": 0}
        for i, g in enumerate(grads):
            slot = gbuf.view(i)
            if g is None:
                slot.zero_()
                gbuf.stats["zeroed"] += 1
            elif g.data_ptr() == slot.data_ptr() and g.stride() == slot.stride() and g.dtype == slot.dtype:
                gbuf.stats["in_place"] += 1
            else:
                slot.copy_(g)                           # a gradient that was not written in place (fallback)
                gbuf.stats["copied"] += 1
        dkv = gbuf.flat()
        gbuf.release()
        return dkv, None, None, None


def split_kv(kv, n_layers: int, n_heads: int):
    """kv [B, Kn, n_layers*2*d] (the c_attn outputs of n_layers layers side by side; n_layers = 1: one layer's
    c_attn output) -> list of (key, value, slot) per layer: key / value are [B, h, Kn, hd] views, slot is the
    KVGradSlot to pass as xattn(kv_grad=slot) so that the layer's dK / dV are written straight into the one gradient
    tensor that flows back into the projection.  Pure torch apart from that hand-over: runs on any device."""
    if kv.dim() != 3 or kv.shape[2] % (2 * n_layers * n_heads):
        raise ValueError(f"split_kv: kv {tuple(kv.shape)}: want [B, Kn, {n_layers}*2*d] with d divisible by {n_heads}")
    if not kv.is_contiguous():
        kv = kv.contiguous()
    gbuf = _KVGradBuffer(kv.shape, kv.dtype, kv.device, int(n_layers), int(n_heads))
    outs = _SplitKV.apply(kv, int(n_layers), int(n_heads), gbuf)
    return [(outs[2 * l], outs[2 * l + 1], KVGradSlot(gbuf, l)) for l in range(n_layers)]


def colsum(x2d):
    """Column sums of a [M, N] bf16 / fp32 matrix in fp32, fixed summation order (star_colsum): the bias gradient of a
    projection."""
    _need_cuda(x2d)
    if x2d.dim() != 2 or x2d.dtype not in (torch.bfloat16, torch.float32):
        raise TypeError(f"colsum: want a 2-D bf16 / fp32 tensor, got {tuple(x2d.shape)} {x2d.dtype}")
    if x2d.stride(1) != 1:
        x2d = x2d.contiguous()
    M, N = x2d.shape
    out = torch.empty((N,), dtype=torch.float32, device=x2d.device)
    wsb = lib().star_colsum_workspace(M, N)
    ws = torch.empty((wsb,), dtype=torch.uint8, device=x2d.device)
    check(lib().star_colsum(ptr(x2d), DT_BF16 if x2d.dtype == torch.bfloat16 else DT_F32, M, N, x2d.stride(0), ptr(out),
                            ptr(ws), wsb, stream_ptr()), "star_colsum")
    return out


def transpose_cat_bf16(weights):
    """[w_i [in, out] for i] (contiguous, all fp32 or all bf16, one shape) -> bf16 [len * out, in]: every weight
    transposed, side by side, in one launch (star_transpose_list_bf16)."""
    n = len(weights)
    _need_cuda(*weights)
    w0 = weights[0]
    if n < 1 or n > TN_MAX_BATCH or w0.dim() != 2 or w0.dtype not in (torch.bfloat16, torch.float32):
        raise ValueError(f"transpose_cat_bf16: 1..{TN_MAX_BATCH} 2-D bf16 / fp32 matrices")
    for w in weights:
        if w.shape != w0.shape or w.dtype != w0.dtype or not w.is_contiguous() or w.device != w0.device:
            raise ValueError("transpose_cat_bf16: the matrices must be contiguous and share one shape, dtype and device")
    rows, cols = w0.shape
    out = torch.empty((n * cols, rows), dtype=torch.bfloat16, device=w0.device)
    pw = (C.c_void_p * n)(*[w.data_ptr() for w in weights])
    check(lib().star_transpose_list_bf16(pw, n, DT_BF16 if w0.dtype == torch.bfloat16 else DT_F32, rows, cols, ptr(out),
                                         stream_ptr()), "star_transpose_list_bf16")
    return out


def colsum_batched(xs):
    """star_colsum of up to 16 matrices of one shape, dtype and row stride in one pair of launches -> fp32 [len(xs), N]
    (each row bit-identical to colsum of that matrix)."""
    nb = len(xs)
    _need_cuda(*xs)
    x0 = xs[0]
    if nb < 1 or nb > TN_MAX_BATCH or x0.dim() != 2 or x0.dtype not in (torch.bfloat16, torch.float32):
        raise ValueError(f"colsum_batched: 1..{TN_MAX_BATCH} 2-D bf16 / fp32 matrices")
    for x in xs:
        if x.shape != x0.shape or x.dtype != x0.dtype or x.stride() != x0.stride() or x.stride(1) != 1 \
                or x.device != x0.device:
            raise ValueError("colsum_batched: the matrices must share one shape, dtype, row stride and device")
    M, N = x0.shape
    out = torch.empty((nb, N), dtype=torch.float32, device=x0.device)
    wsb = nb * lib().star_colsum_workspace(M, N)
    ws = torch.empty((wsb,), dtype=torch.uint8, device=x0.device)
    px = (C.c_void_p * nb)(*[x.data_ptr() for x in xs])
    check(lib().star_colsum_batched(px, nb, DT_BF16 if x0.dtype == torch.bfloat16 else DT_F32, M, N, x0.stride(0),
                                    ptr(out), ptr(ws), wsb, stream_ptr()), "star_colsum_batched")
    return out


def gemm_nt_bf16(a2d, wt, bias=None):
    """out[M,N] = a2d[M,K] @ wt[N,K]^T + bias[N] on this library's 256 x 256 MFMA tile (star_gemm_nt_bf16)."""
    _need_cuda(a2d, wt, bias)
    if a2d.dtype != torch.bfloat16 or wt.dtype != torch.bfloat16:
        raise TypeError("gemm_nt_bf16: bf16 operands only")
    if a2d.stride(1) != 1:
        a2d = a2d.contiguous()
    if wt.stride(1) != 1:
        wt = wt.contiguous()
    M, K = a2d.shape
    N = wt.shape[0]
    if wt.shape[1] != K:
        raise ValueError(f"gemm_nt_bf16: a {tuple(a2d.shape)} x wt {tuple(wt.shape)}^T")
    bf = None if bias is None else bias.detach().to(torch.float32).contiguous()
    out = torch.empty((M, N), dtype=torch.bfloat16, device=a2d.device)
    check(lib().star_gemm_nt_bf16(ptr(a2d), a2d.stride(0), ptr(wt), wt.stride(0), ptr(bf), M, N, K, ptr(out), N,
                                  stream_ptr()), "star_gemm_nt_bf16")
    return out


TN_MAX_BATCH = 16            # star_gemm_tn_batched_bf16: products (times r ranges) per launch
_TN_MAX_ROWS = 7168          # rows of one r range


def gemm_tn_batched(a_list, b_list, ksplit: int = 1, reduce: bool = True):
    """[a_i^T @ b_i for i] in one launch of this library's TN GEMM (star_gemm_tn_batched_bf16): a_i [R, M], b_i [R, N]
    bf16, all of one shape and row stride -> fp32 [len(a_list), M, N].  ksplit: r ranges per product (their partial
    products are added here, in order; reduce=False returns them as they are, [len(a_list), ksplit, M, N])."""
    nb = len(a_list)
    if nb < 1 or len(b_list) != nb or nb * ksplit > TN_MAX_BATCH:
        raise ValueError(f"gemm_tn_batched: {nb} products x {ksplit} ranges (at most {TN_MAX_BATCH} together)")
    _need_cuda(*a_list, *b_list)
    a0, b0 = a_list[0], b_list[0]
    for a, b in zip(a_list, b_list):
        if a.dtype != torch.bfloat16 or b.dtype != torch.bfloat16 or a.dim() != 2 or b.dim() != 2:
            raise TypeError("gemm_tn_batched: 2-D bf16 operands only")
        if a.shape != a0.shape or b.shape != b0.shape or a.stride() != a0.stride() or b.stride() != b0.stride() \
                or a.shape[0] != b.shape[0] or a.stride(1) != 1 or b.stride(1) != 1 or a.device != a0.device:
            raise ValueError("gemm_tn_batched: the products must share one shape, row stride and device")
    R, M = a0.shape
    N = b0.shape[1]
    out = torch.empty((nb * ksplit, M, N), dtype=torch.float32, device=a0.device)
    pa = (C.c_void_p * nb)(*[a.data_ptr() for a in a_list])
    pb = (C.c_void_p * nb)(*[b.data_ptr() for b in b_list])
    check(lib().star_gemm_tn_batched_bf16(pa, pb, nb, ksplit, R, M, N, a0.stride(0), b0.stride(0), ptr(out),
                                          stream_ptr()), "star_gemm_tn_batched_bf16")
    if not reduce:
        return out.view(nb, ksplit, M, N)
    return out if ksplit == 1 else out.view(nb, ksplit, M, N).sum(1)


def _tn_batch_fits(x2, g2) -> bool:
    R, M = x2.shape
    N = g2.shape[1]
    ks = 1
    while R // ks > _TN_MAX_ROWS and ks < TN_MAX_BATCH:
        ks *= 2
    return (x2.is_cuda and x2.dtype == torch.bfloat16 and g2.dtype == torch.bfloat16 and R % (64 * ks) == 0
            and R // ks <= _TN_MAX_ROWS and M >= 128 and N >= 128 and M % 8 == 0 and N % 8 == 0
            and x2.stride(1) == 1 and g2.stride(1) == 1 and x2.stride(0) % 8 == 0 and g2.stride(0) % 8 == 0
            and x2.data_ptr() % 16 == 0 and g2.data_ptr() % 16 == 0)


class _DeferredWeightGrads:
    """Weight gradients of the small projections (q_attn / c_proj: dW = x^T g, [d, B*L] x [B*L, d]), held back until
    the backward pass ends and then formed together.  One of them has 16 output tiles of 256 x 256 -- hipBLASLt gives
    it 64 x 64 tiles at 0.28 PFLOP/s, 31 us at M = 4096, d = 1024 -- while the eight of a 4-layer decoder fill the chip
    in one launch of star_gemm_tn_batched_bf16.  Opt-in (Conv1D.defer_weight_grad / defer_projection_weight_grads):
    the gradients are written to `.grad` by an end-of-backward callback of the autograd engine, so they do not exist
    for torch.autograd.grad() callers and gradient hooks on the weights (hook-based DDP) do not see them."""

    def __init__(self):
        self.items = []              # (x2, g2, weight, bias or None)
        self.armed = False

    def add(self, x2, g2, weight, bias=None):
        self.items.append((x2, g2, weight, bias))
        if not self.armed:
            self.armed = True
            torch.autograd.Variable._execution_engine.queue_callback(self.flush)

    def flush(self):
        items, self.items, self.armed = self.items, [], False
        groups = {}
        for x2, g2, w, b in items:
            groups.setdefault((tuple(x2.shape), x2.stride(0), tuple(g2.shape), g2.stride(0), x2.device), []).append(
                (x2, g2, w, b))
        with torch.no_grad():
            for grp in groups.values():
                R, M = grp[0][0].shape
                N = grp[0][1].shape[1]
                tiles = ((M + 255) // 256) * ((N + 255) // 256)
                # r ranges per product: as many (a power of two) as it takes to reach one workgroup per CU, within the
                # 16 (product, range) pairs of a launch -- 8 products of 1024 x 1024: 2 ranges, 256 workgroups
                ks = 1
                while (tiles * min(len(grp), TN_MAX_BATCH // ks) * ks < 256 and ks < TN_MAX_BATCH
                       and R % (128 * ks) == 0) or R // ks > _TN_MAX_ROWS:
                    ks *= 2
                per = max(1, TN_MAX_BATCH // ks)
                for at in range(0, len(grp), per):
                    part = grp[at:at + per]
                    raw = gemm_tn_batched([it[0] for it in part], [it[1] for it in part], ks, reduce=False)
                    wdt = {it[2].dtype for it in part}
                    if ks == 2 and len(wdt) == 1:
                        # the sum of the two ranges and the cast to the weights' dtype in one pass over the partials
                        # (one fp32 add, one rounding: the same value as sum-then-cast)
                        dws = torch.add(raw[:, 0], raw[:, 1], out=torch.empty(raw.shape[0], M, N, dtype=wdt.pop(),
                                                                              device=raw.device))
                    else:
                        dws = raw[:, 0] if ks == 1 else raw.sum(1)
                    with_bias = [it for it in part if it[3] is not None]
                    dbs = colsum_batched([it[1] for it in with_bias]) if with_bias else None
                    by_dtype = {}

                    def cast(t, dtype, tag):
                        if (tag, dtype) not in by_dtype:
                            by_dtype[(tag, dtype)] = t if dtype == t.dtype else t.to(dtype)
                        return by_dtype[(tag, dtype)]

                    nbias = 0
                    for i, (_, _, w, b) in enumerate(part):
                        grads = [(w, cast(dws, w.dtype, "w")[i])]
                        if b is not None:
                            grads.append((b, cast(dbs, b.dtype, "b")[nbias]))
                            nbias += 1
                        for prm, gr in grads:
                            if prm.grad is None:
                                prm.grad = gr
                            else:
                                prm.grad.add_(gr)


_deferred_wgrads = _DeferredWeightGrads()


def _own_gemm_fits(M: int, N: int, K: int, dtype) -> bool:
    """The 256 x 256 tile fills the chip from ~256 output tiles on (MI355X: 256 CUs); below that hipBLASLt's smaller
    tiles win."""
    return dtype == torch.bfloat16 and N % 64 == 0 and K % 64 == 0 and ((M + 255) // 256) * ((N + 255) // 256) >= 192


class _Conv1D(torch.autograd.Function):
    """HF Conv1D (y = x @ W + b, W [in, out]; the projections around the core, my_transformer.py:500-501,526).
    Forward: this library's GEMM with the bias in its epilogue when the output has enough 256 x 256 tiles to fill the
    chip (given wt = W^T in [out, in] layout), else torch.addmm (hipBLASLt).  Backward: two GEMMs through torch, the
    bias gradient by star_colsum (torch's column reduction of a [4096, 1024] bf16 gradient takes 19-22 us, this 5)."""

    @staticmethod
    def forward(ctx, x, weight, bias, wt, defer_wgrad=False):
        ctx.set_materialize_grads(False)
        ctx.defer_wgrad = bool(defer_wgrad)
        x2 = x.reshape(-1, x.shape[-1])
        M, K = x2.shape
        N = weight.shape[1]
        if wt is not None and x2.is_cuda and _own_gemm_fits(M, N, K, x2.dtype):
            y = gemm_nt_bf16(x2, wt, bias)
        else:
            y = torch.addmm(bias.to(x2.dtype), x2, weight.to(x2.dtype))
        ctx.save_for_backward(x2, weight)
        ctx.x_shape = x.shape
        ctx.b_dtype = bias.dtype
        ctx.bias_ref = bias if ctx.defer_wgrad else None
        return y.view(x.shape[:-1] + (N,))

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return None, None, None, None, None
        x2, weight = ctx.saved_tensors
        g2 = g.reshape(-1, g.shape[-1])
        if g2.stride(1) != 1 or g2.stride(0) != g2.shape[1]:
            g2 = g2.contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.mm(g2, weight.to(g2.dtype).t()).view(ctx.x_shape)
        if ctx.needs_input_grad[1]:
            if ctx.defer_wgrad and weight.is_leaf and _tn_batch_fits(x2, g2):
                # formed with the other layers' at the end of backward; the bias gradient (a column sum of the same
                # g) goes with it when the bias is a leaf too
                b = ctx.bias_ref if (ctx.needs_input_grad[2] and ctx.bias_ref is not None and ctx.bias_ref.is_leaf) else None
                _deferred_wgrads.add(x2, g2, weight, b)
                if b is not None:
                    return dx, None, None, None, None
            else:
                dw = torch.mm(x2.t(), g2).to(weight.dtype)
        if ctx.needs_input_grad[2]:
            if g2.is_cuda and g2.dtype in (torch.bfloat16, torch.float32) and g2.shape[1] % 8 == 0:
                db = colsum(g2).to(ctx.b_dtype)
            else:
                db = g2.sum(0).to(ctx.b_dtype)
        return dx, dw, db, None, None


class _SharedConv1D(torch.autograd.Function):
    """The c_attn projections of nl decoder layers on one input (the encoder memory) as ONE GEMM, without the
    per-step concatenation of the layers' weights: forward out = x @ wt_all^T + b_all on this library's GEMM (wt_all
    [nl*2d, d] = the weights transposed side by side, kept by the caller while the weights do not change); backward
    dx = g @ wt_all (no transpose needed: wt_all IS the [K, N] operand), dW_all = x^T @ g handed back to every layer's
    weight as a column slice, the bias gradients by star_colsum."""

    @staticmethod
    def forward(ctx, x2, wt_all, b_all, n_each, *params):
        ctx.set_materialize_grads(False)
        M, K = x2.shape
        N = wt_all.shape[0]
        if _own_gemm_fits(M, N, K, x2.dtype):
            y = gemm_nt_bf16(x2, wt_all, b_all)
        else:
            y = torch.addmm(b_all.to(x2.dtype), x2, wt_all.t())
        ctx.save_for_backward(x2, wt_all)
        ctx.n_each = n_each
        ctx.p_dtypes = [p.dtype for p in params]
        return y

    @staticmethod
    def backward(ctx, g):
        nl = len(ctx.p_dtypes) // 2
        if g is None:
            return (None,) * (4 + 2 * nl)
        x2, wt_all = ctx.saved_tensors
        g2 = g if (g.stride(1) == 1 and g.stride(0) == g.shape[1]) else g.contiguous()
        dx = torch.mm(g2, wt_all) if ctx.needs_input_grad[0] else None
        dws = dbs = [None] * nl
        if any(ctx.needs_input_grad[4:4 + nl]):
            dw_all = torch.mm(x2.t(), g2)
            dws = [dw_all[:, l * ctx.n_each:(l + 1) * ctx.n_each].to(ctx.p_dtypes[l]) for l in range(nl)]
        if any(ctx.needs_input_grad[4 + nl:]):
            db_all = colsum(g2)
            dbs = [db_all[l * ctx.n_each:(l + 1) * ctx.n_each].to(ctx.p_dtypes[nl + l]) for l in range(nl)]
        return (dx, None, None, None, *dws, *dbs)


def shared_conv1d(x2, wt_all, b_all, weights, biases):
    """x2 [M, d] bf16 -> [M, nl * n_each]: see _SharedConv1D.  weights / biases: the layers' Conv1D parameters (the
    gradients go to them); wt_all / b_all: their transposed / concatenated copies."""
    _need_cuda(x2, wt_all, b_all)
    n_each = weights[0].shape[1]
    return _SharedConv1D.apply(x2, wt_all, b_all, n_each, *weights, *biases)


def conv1d(x, weight, bias, wt=None, defer_weight_grad=False):
    """y = x @ weight + bias for an HF Conv1D (weight [in, out]); wt: optional bf16 copy of weight^T ([out, in]) that
    lets wide projections run on star_gemm_nt_bf16.  defer_weight_grad: see _DeferredWeightGrads (opt-in: the weight's
    gradient is written to .grad when the backward pass ends, batched with the other deferred projections)."""
    return _Conv1D.apply(x, weight, bias, wt, defer_weight_grad)


def share_kv(mem, w_all, b_all, n_layers: int, n_heads: int, wt_all=None):
    """K / V of n_layers cross-attention layers from ONE projection of the encoder memory.

    mem [B, Kn, d]; w_all [d, n_layers*2*d] = the layers' c_attn weights (HF Conv1D layout [in, out]) side by side,
    b_all [n_layers*2*d]; wt_all (optional, bf16 [n_layers*2*d, d]) = w_all^T, which puts the forward on
    star_gemm_nt_bf16 with the bias in its epilogue.  One GEMM forward; backward dmem = dKV W^T and dW = mem^T dKV over
    all layers at once, the bias gradient by star_colsum.  Returns split_kv() of the projection output."""
    if mem.dim() != 3 or w_all.dim() != 2 or w_all.shape[0] != mem.shape[2]:
        raise ValueError(f"share_kv: mem {tuple(mem.shape)} / w_all {tuple(w_all.shape)}: want [B,Kn,d] and [d, nl*2*d]")
    d = mem.shape[2]
    if w_all.shape[1] != n_layers * 2 * d or b_all.shape != (n_layers * 2 * d,):
        raise ValueError(f"share_kv: w_all / b_all must have {n_layers}*2*{d} columns, got {tuple(w_all.shape)} / "
                         f"{tuple(b_all.shape)}")
    if d % n_heads:
        raise ValueError("share_kv: d must be divisible by n_heads")
    if not (mem.dtype == w_all.dtype == b_all.dtype):
        raise TypeError("share_kv: mem, w_all and b_all must share one dtype")
    B, Kn, _ = mem.shape
    kv = conv1d(mem.reshape(B * Kn, d), w_all, b_all, wt_all).view(B, Kn, -1)
    return split_kv(kv, n_layers, n_heads)


def xattn(q, k, v, bias=None, key_len=None, ilk_diag: int = -1, scale: Optional[float] = None,
          need_probs: bool = False, dropout_p: float = 0.0, dropout_rng=None, kv_grad=None):
    """q [B,h,L,hd]; k, v [B,h,Kn,hd]; bias [B,bias_len<=Kn] fp32 added to every query's logits (keys past
    bias_len get 0: the prefix-bias variant, my_transformer.py:114-125); key_len [B] masks keys >= key_len[b];
    ilk_diag >= 0 enables the infinite-lookback mask (key j visible iff j <= i+diag).
    dropout_p > 0: attention-probability dropout (my_transformer.py:595) with the counter-based mask specified in
    star_nugget.h; dropout_rng = (seed, offset) host integers for a reproducible mask, a device int64[2] tensor, or
    None to draw from the per-device state seeded by torch's generator.
    kv_grad: the KVGradSlot that share_kv() returned with this layer's k and v (optional; see there).
    Returns (O [B,h,L,hd] (memory order [B,L,h,hd]), P or None); P holds the dropped, rescaled weights."""
    _need_cuda(q, k, v, bias, key_len)
    if scale is None:
        scale = 1.0 / math.sqrt(q.shape[-1])
    if not 0.0 <= dropout_p < 1.0:
        raise ValueError(f"xattn: dropout_p = {dropout_p} not in [0, 1)")
    return _XAttn.apply(q, k, v, bias, key_len, int(ilk_diag), float(scale), bool(need_probs), float(dropout_p),
                        dropout_rng, kv_grad)


def test_override(which: int, value: int) -> int:
    """TEST-ONLY: force a kernel variant (star_test_override in star_nugget.h); returns the previous value."""
    rc = lib().star_test_override(int(which), int(value))
    if rc < 0:
        check(rc, "star_test_override")
    return rc


OVR_XA_BWD, OVR_XA_DQ, OVR_XA_DKV, OVR_XA_NG, OVR_XA_FWD, OVR_SCORER_NT, OVR_SCORER_FWD = 0, 1, 2, 3, 4, 5, 6
