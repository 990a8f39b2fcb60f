"""Streaming inference side of the nugget path: incremental scorer, online fire decision
and the read/write policy that the infinite-lookback mask implies at generation time.

The reference trains the simultaneous model with whole utterances (w2v_simul.py:407-465: scorer -> sigmoid ->
resize -> apply_threshold_yield_nugget -> gathers -> decoder with limit_cross_attn) and has no chunked inference loop;
what it fixes is the arithmetic: per-frame scorer (w2v_simul.py:410-411), alphas = sigmoid(score) * mask (:420-421),
the fp32 integrate-and-fire recurrence (:376-403), the row / score gathers at the fire positions (:427, :446-447) and
the mask "token i sees nuggets j <= i + 2" (my_transformer.py:548-556).  This module feeds the same kernels one chunk
at a time:

  * scorer: a per-frame MLP, so the chunk is scored by the same star_scorer_fwd launch on [B*Tc, d];
  * fire decision: star_cif_fire_step carries the integrator across chunks; any chunking gives the fires of the
    whole-utterance scan bit for bit (tests/test_gpu_round2.py);
  * memory: the fired rows of the chunk's semantic states are appended to a preallocated [B, Kmax, d] memory with
    their scorer logits (the attention bias), without a host sync;
  * policy: with m nuggets read, tokens 0 .. m-3 see exactly the keys they would see offline (keys j <= i + 2), so
    they may be written; once the utterance has ended every token may be.

`resize` needs the target length and is a training-time device (w2v_simul.py:425); it is not part of this path.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import ops

INT32_MAX = 2 ** 31 - 1


class StreamingNuggetSegmenter:
    """Online nugget segmenter for a batch of B concurrent utterances.

        seg = StreamingNuggetSegmenter(scorer, batch=B, dim=d, max_nuggets=Kmax, device=dev)
        for x_chunk, s_chunk in stream:                  # [B,Tc,d] encoder / semantic states of the next Tc frames
            seg.push(x_chunk, s_chunk, lengths=None)     # lengths: [B] int32 absolute utterance lengths once known
            n_ready = seg.tokens_ready()                 # [B] tokens that may be written now
        mem, scores, counts = seg.memory()

    Nothing here synchronises with the host; counts live on the device."""

    def __init__(self, scorer, batch: int, dim: int, max_nuggets: int, device, lookahead: int = 2,
                 dtype=torch.bfloat16):
        self.scorer = scorer
        self.B, self.d, self.kmax = int(batch), int(dim), int(max_nuggets)
        self.lookahead = int(lookahead)                 # ILK: token i sees nuggets j <= i + lookahead
        self.state = ops.FireState(self.B, device)
        # slot kmax is a dump slot: the pad entries of a chunk's fire list are scattered there
        self.mem = torch.zeros(self.B, self.kmax + 1, self.d, dtype=dtype, device=device)
        self.mem_scores = torch.zeros(self.B, self.kmax + 1, dtype=torch.float32, device=device)
        self.mem_pos = torch.full((self.B, self.kmax + 1), -1, dtype=torch.int64, device=device)
        self.counts = torch.zeros(self.B, dtype=torch.int32, device=device)
        self.lengths = torch.full((self.B,), INT32_MAX, dtype=torch.int32, device=device)
        self._zeros = torch.zeros(self.B, dtype=torch.int32, device=device)

    def reset(self, rows: Optional[torch.Tensor] = None):
        """Start new utterances in the given batch rows (bool [B] or index tensor; default all)."""
        sel = slice(None) if rows is None else rows
        self.state.cur[sel] = 0
        self.state.pos[sel] = 0
        self.counts[sel] = 0
        self.lengths[sel] = INT32_MAX
        self.mem_pos[sel] = -1

    @torch.no_grad()
    def push(self, x_chunk: torch.Tensor, s_chunk: torch.Tensor, lengths: Optional[torch.Tensor] = None,
             cap: Optional[int] = None):
        """Score the chunk, decide its fires, append the fired semantic rows and their logits to the memory.
        Returns (idx [B,cap] absolute fire positions of this chunk, -1 padded; counts [B] fires of this chunk)."""
        B, Tc, d = x_chunk.shape
        if (B, d) != (self.B, self.d) or s_chunk.shape != x_chunk.shape:
            raise ValueError(f"push: chunk {tuple(x_chunk.shape)} / {tuple(s_chunk.shape)} does not match B={self.B}, d={self.d}")
        if lengths is not None:
            self.lengths.copy_(lengths.to(torch.int32))
        cap = Tc if cap is None else int(cap)
        scores = self.scorer(x_chunk).squeeze(-1).float()                       # w2v_simul.py:410-411
        rel = (self.lengths.to(torch.int64) - self.state.pos.to(torch.int64)).clamp_(0, Tc).to(torch.int32)
        alphas, _ = ops.cif_quantity(scores, self._zeros, lengths=rel)          # sigmoid * mask, :420-421
        pos0 = self.state.pos.to(torch.int64)
        idx, cnt, sg = ops.cif_fire_step(alphas, self.lengths, self.state, cap=cap, scores=scores)
        # rows of the chunk at the fire positions (chunk-relative; a pad entry reads the chunk's last frame)
        local = torch.where(idx >= 0, idx - pos0[:, None], torch.full_like(idx, -1))
        rows = ops.gather_rows(s_chunk, local)                                   # [B,cap,d], :446-447
        j = torch.arange(cap, device=idx.device)[None, :]
        dest = self.counts.to(torch.int64)[:, None] + j
        ok = (j < cnt[:, None]) & (dest < self.kmax)
        dest = torch.where(ok, dest, torch.full_like(dest, self.kmax))
        self.mem.scatter_(1, dest[:, :, None].expand(-1, -1, d), rows.to(self.mem.dtype))
        self.mem_scores.scatter_(1, dest, sg)
        self.mem_pos.scatter_(1, dest, idx)
        self.counts.add_(cnt).clamp_(max=self.kmax)
        return idx, cnt

    def memory(self):
        """(mem [B,Kmax,d], scorer logits [B,Kmax] = the attention bias, fire positions [B,Kmax], counts [B] int32)."""
        return self.mem[:, :self.kmax], self.mem_scores[:, :self.kmax], self.mem_pos[:, :self.kmax], self.counts

    def finished(self):
        """[B] bool: every frame of the utterance has been consumed."""
        return self.state.pos >= self.lengths

    def tokens_ready(self):
        """[B] int64: how many target tokens may be written now.  Token i attends nuggets j <= i + lookahead
        (my_transformer.py:548-556), so with m nuggets read tokens 0 .. m - lookahead - 1 already see their final
        key set; after the end of the utterance there is no limit (INT32_MAX)."""
        ready = (self.counts.to(torch.int64) - self.lookahead).clamp_(min=0)
        return torch.where(self.finished(), torch.full_like(ready, INT32_MAX), ready)

    def ilk_diag_for(self, token_index: int) -> int:
        """ilk_diag argument of ops.xattn for a single-query call that decodes target token `token_index`: the
        kernel's predicate is j <= i_local + diag with i_local = 0 for the only query row."""
        return int(token_index) + self.lookahead
