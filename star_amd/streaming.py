"""Streaming inference side of the nugget path: incremental scorer, online fire decision,
the growing nugget memory with its K/V projections, and the read/write policy that the infinite-lookback mask implies
at generation time.

The reference trains the simultaneous model with whole utterances (w2v_simul.py:407-465: scorer -> sigmoid ->
resize -> apply_threshold_yield_nugget -> gathers -> decoder with limit_cross_attn) and has no chunked inference loop;
what it fixes is the arithmetic: per-frame scorer (w2v_simul.py:410-411), alphas = sigmoid(score) * mask (:420-421),
the fp32 integrate-and-fire recurrence (:376-403), the row / score gathers at the fire positions (:427, :446-447), the
mask "token i sees nuggets j <= i + 2" (my_transformer.py:563-570) and the per-token decoder call that re-projects the
unchanged memory for every token (my_transformer.py:724-733, 1065-1100).  This module feeds the same kernels one chunk
at a time, in TWO launches per chunk plus the scorer's finish kernel:

  * scorer: a per-frame MLP, so the chunk is scored by the same star_scorer_fwd launch on [B*Tc, d] (bf16 weight
    copies are made once, not per chunk);
  * star_cif_stream_push: sigmoid * mask, the fire decision with the integrator carried across chunks (any chunking
    gives the fires of the whole-utterance scan bit for bit) and the append of the fired semantic rows, logits and
    positions to a preallocated [B, Kmax, d] memory, without a host sync and without an allocation;
  * policy: with m nuggets read, tokens 0 .. m-3 see exactly the keys they would see offline (keys j <= i + 2), so
    they may be written; once the utterance has ended every token may be;
  * StreamingEncoderKV: the decoder layers' K / V of the memory, projected incrementally (only the rows appended since
    the last call), handed to NuggetCrossAttention as `encoder_kv=` for token-by-token decoding.

Limits: at most `max_nuggets` nuggets per utterance -- fires beyond that are dropped and COUNTED (`overflow()`; poll
it, a non-zero entry means that utterance's memory is truncated); the end of an utterance must be announced no later
than the push after the one that delivered its last frame (or by `flush`), see star_cif_stream_push.

`resize` needs the target length and is a training-time device (w2v_simul.py:425); it is not part of this path.
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch

from . import ops
from ._lib import check, lib, ptr, stream_ptr

INT32_MAX = 2 ** 31 - 1


class StreamingNuggetSegmenter:
    """Online nugget segmenter for a batch of B concurrent utterances.

        seg = StreamingNuggetSegmenter(scorer, batch=B, dim=d, max_nuggets=Kmax, device=dev)
        for x_chunk, s_chunk in stream:                  # [B,Tc,d] encoder / semantic states of the next Tc frames
            seg.push(x_chunk, s_chunk, lengths=None)     # lengths: [B] int32 absolute utterance lengths once known
            n_ready = seg.tokens_ready()                 # [B] tokens that may be written now
        seg.flush(lengths)                               # only if the end was not announced with the last chunk
        mem, scores, positions, counts = seg.memory()

    Nothing here synchronises with the host or allocates after the first push of a chunk size; counts live on the
    device.  The scorer's weights are copied to bf16 at construction: call refresh_weights() after changing them."""

    def __init__(self, scorer, batch: int, dim: int, max_nuggets: int, device, lookahead: int = 2,
                 dtype=torch.bfloat16):
        self.scorer = scorer
        self.B, self.d, self.kmax = int(batch), int(dim), int(max_nuggets)
        self.lookahead = int(lookahead)                 # ILK: token i sees nuggets j <= i + lookahead
        self.device = torch.device(device)
        self.state = ops.StreamState(self.B, self.d, self.kmax, self.device, dtype)
        self.lengths = torch.full((self.B,), INT32_MAX, dtype=torch.int32, device=self.device)
        self._bufs = {}                                 # chunk size -> (scores, workspace, idx, cnt)
        self.refresh_weights()

    @torch.no_grad()
    def refresh_weights(self):
        """bf16 copy of W1 (nn.Linear layout) and fp32 copies of b1, w2, b2: what star_scorer_fwd consumes."""
        sc = self.scorer
        self._w1b = sc[0].weight.detach().to(self.device, torch.bfloat16).contiguous()
        self._b1 = sc[0].bias.detach().to(self.device, torch.float32).contiguous()
        self._w2 = sc[2].weight.detach().to(self.device, torch.float32).contiguous().view(-1)
        self._b2 = sc[2].bias.detach().to(self.device, torch.float32).contiguous().view(-1)

    def reset(self, rows: Optional[torch.Tensor] = None):
        """Start new utterances in the given batch rows (bool [B] or index tensor; default all)."""
        self.state.reset(rows)
        self.lengths[slice(None) if rows is None else rows] = INT32_MAX

    def _buffers(self, Tc: int):
        ent = self._bufs.get(Tc)
        if ent is None:
            N = self.B * Tc
            wsb = lib().star_scorer_fwd_workspace(N, self.d)
            ent = (torch.empty((self.B, Tc), dtype=torch.float32, device=self.device),
                   torch.empty((wsb,), dtype=torch.uint8, device=self.device), wsb,
                   torch.empty((self.B, Tc), dtype=torch.int64, device=self.device),
                   torch.empty((self.B,), dtype=torch.int32, device=self.device))
            self._bufs[Tc] = ent
        return ent

    @torch.no_grad()
    def push(self, x_chunk: torch.Tensor, s_chunk: torch.Tensor, lengths: Optional[torch.Tensor] = None):
        """Score the chunk, decide its fires, append the fired semantic rows and their logits to the memory.
        Returns (idx [B,Tc] absolute fire positions of this chunk, -1 padded; counts [B] fires of this chunk) --
        views of buffers that the next push of the same chunk size overwrites."""
        B, Tc, d = x_chunk.shape
        if (B, d) != (self.B, self.d) or s_chunk.shape != x_chunk.shape:
            raise ValueError(f"push: chunk {tuple(x_chunk.shape)} / {tuple(s_chunk.shape)} does not match B={self.B}, d={self.d}")
        ops._need_cuda(x_chunk, s_chunk, lengths)
        if lengths is not None:
            self.lengths.copy_(lengths)
        scores, ws, wsb, idx, cnt = self._buffers(Tc)
        xb = x_chunk if x_chunk.dtype == torch.bfloat16 else x_chunk.to(torch.bfloat16)
        xb = xb.contiguous()
        # w2v_simul.py:410-411: the per-frame scorer on the chunk's frames (inference: no ReLU sign bits kept)
        check(lib().star_scorer_fwd(ptr(xb), ptr(self._w1b), ptr(self._b1), ptr(self._w2), ptr(self._b2), B * Tc, d,
                                    ptr(scores), None, ptr(ws), wsb, stream_ptr()), "star_scorer_fwd")
        sc = s_chunk if s_chunk.dtype == self.state.dtype else s_chunk.to(self.state.dtype)
        ops.cif_stream_push(scores, sc.contiguous(), self.lengths, self.state, idx_out=idx, cnt_out=cnt)
        return idx, cnt

    @torch.no_grad()
    def flush(self, lengths: torch.Tensor):
        """The utterance lengths became known after the last frames were pushed (a live stream that simply stops):
        emits the forced last-frame fire (w2v_simul.py:388) of every utterance whose length equals the frames consumed
        and whose last frame did not fire by itself."""
        self.lengths.copy_(lengths)
        ops.cif_stream_push(None, None, self.lengths, self.state)

    def memory(self):
        """(mem [B,Kmax,d], scorer logits [B,Kmax] = the attention bias, fire positions [B,Kmax], counts [B] int32)."""
        st = self.state
        return st.mem, st.mem_scores, st.mem_pos, st.counts

    def overflow(self):
        """[B] int32: fires that found no slot in the memory (max_nuggets too small).  Non-zero = truncated memory."""
        return self.state.dropped

    def finished(self):
        """[B] bool: every frame of the utterance has been consumed."""
        return self.state.pos >= self.lengths

    def tokens_ready(self):
        """[B] int64: how many target tokens may be written now.  Token i attends nuggets j <= i + lookahead
        (my_transformer.py:563-570), so with m nuggets read tokens 0 .. m - lookahead - 1 already see their final
        key set; after the end of the utterance there is no limit (INT32_MAX)."""
        ready = (self.state.counts.to(torch.int64) - self.lookahead).clamp_(min=0)
        return torch.where(self.finished(), torch.full_like(ready, INT32_MAX), ready)

    def ilk_diag_for(self, token_index: int) -> int:
        """ilk_diag argument of ops.xattn for a single-query call that decodes target token `token_index`: the
        kernel's predicate is j <= i_local + diag with i_local = 0 for the only query row."""
        return int(token_index) + self.lookahead


class StreamingEncoderKV:
    """K / V of the growing nugget memory for every decoder layer's cross-attention, projected incrementally.

    The reference projects the whole memory again in every block for every generated token (`c_attn(
    encoder_hidden_states)`, my_transformer.py:500-501 under 724-733 / 1079-1099).  c_attn is row-wise, so a row that
    entered the memory is projected ONCE, for all layers at once ([rows, d] x [d, nl*2d]), into a preallocated
    [B, Kmax, nl*2d] buffer; extend(counts) projects the rows appended since the last call.  layers()
    returns one EncoderKV per layer for NuggetCrossAttention(encoder_kv=...): strided head views, no copies."""

    def __init__(self, attn_modules: Sequence, segmenter: StreamingNuggetSegmenter):
        from .nugget import EncoderKV
        self._EncoderKV = EncoderKV
        mods = list(attn_modules)
        if not mods:
            raise ValueError("StreamingEncoderKV: no attention modules")
        self.seg = segmenter
        self.nl = len(mods)
        self.h = mods[0].num_heads
        dt = segmenter.state.dtype
        with torch.no_grad():
            self.w_all = torch.cat([m.c_attn.weight for m in mods], dim=1).to(segmenter.device, dt).contiguous()
            self.b_all = torch.cat([m.c_attn.bias for m in mods], dim=0).to(segmenter.device, dt).contiguous()
        B, kmax, d = segmenter.B, segmenter.kmax, segmenter.d
        self.kv = torch.zeros(B, kmax, self.nl * 2 * d, dtype=dt, device=segmenter.device)
        self._prev = [0] * B                            # per utterance: rows [0, prev) are projected

    @torch.no_grad()
    def extend(self, counts: Sequence[int]):
        """Project the memory rows appended since the last call, for all layers at once.  counts: the utterances'
        current nugget counts ON THE HOST (the decoding loop reads them anyway to decide what it may write).  The rows
        [min_b previous count, max_b count) of EVERY utterance go through one GEMM: an utterance that is ahead gets a few
        rows projected again (same values), one that is behind gets rows beyond its count projected early (they are
        masked until the count covers them, and projected again then)."""
        cur = [min(int(c), self.seg.kmax) for c in counts]
        if len(cur) != self.seg.B:
            raise ValueError(f"extend: {len(cur)} counts for a batch of {self.seg.B}")
        lo, hi = min(self._prev), max(cur)
        if hi > lo:
            B, d = self.seg.B, self.seg.d
            rows = self.seg.state.mem[:, lo:hi].reshape(-1, d)
            out = torch.addmm(self.b_all, rows, self.w_all)
            self.kv[:, lo:hi] = out.view(B, hi - lo, -1)
        self._prev = cur

    def reset(self, rows: Optional[Sequence[int]] = None):
        for b in (range(self.seg.B) if rows is None else rows):
            self._prev[b] = 0

    def layers(self):
        B, kmax, d, h = self.seg.B, self.seg.kmax, self.seg.d, self.h
        out = []
        for l in range(self.nl):
            k = self.kv[:, :, 2 * l * d:(2 * l + 1) * d].view(B, kmax, h, d // h).permute(0, 2, 1, 3)
            v = self.kv[:, :, (2 * l + 1) * d:(2 * l + 2) * d].view(B, kmax, h, d // h).permute(0, 2, 1, 3)
            out.append(self._EncoderKV(k, v, None))
        return out

    def encoder_attention_mask(self):
        """The additive [B,1,1,Kmax] mask of the memory's live slots (what invert_attention_mask would give the
        reference, my_transformer.py:846-852), from the device-side counts: no host sync."""
        live = torch.arange(self.seg.kmax, device=self.seg.device)[None, :] < self.seg.state.counts[:, None]
        return ((~live).to(torch.float32) * torch.finfo(torch.float32).min)[:, None, None, :]
