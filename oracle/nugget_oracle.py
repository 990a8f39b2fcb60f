"""CPU oracle for the STAR nugget hot path -- TEST INFRASTRUCTURE ONLY.

This file is a from-scratch CPU restatement (numpy for the integer / index work,
torch-CPU fp32 for the differentiable float work) of the reference algorithms
listed below.  It is the *checker* for the HIP kernels in ``star_amd/csrc``.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it; nothing under ``star_amd/`` does.

Parity status: PINNED.  Every function below is checked by
``tests/test_oracle_golden.py`` against golden vectors in ``tests/golden/*.npz``
that were produced by importing the true reference from ``/root/reference`` in
the build container (script: ``tests/golden/gen_golden.py``).

Reference lines restated (all paths relative to the reference checkout):
  scorer                      src/models/w2v_asr.py:113-117,553
  nugget_counts / topk_select src/models/w2v_asr.py:208-244
  type_table                  src/models/w2v_asr.py:316-323
  gather_rows                 src/models/w2v_asr.py:566-567, w2v_simul.py:427,446-447
  quantity_loss               src/models/w2v_simul.py:419-424
  resize                      src/models/w2v_simul.py:504-535
  fire_scan / block mask      src/models/w2v_simul.py:196-220,376-403
  xattn_core                  src/models/my_transformer.py:535-603 (attention dropout :595 with a given mask)
  valid_mask_indices          src/models/w2v_simul.py:196-220
  cif_aggregate               src/models/w2v_simul.py:558-606, w2v_asr.py:681-722  
"""
from __future__ import annotations


import numpy as np
import torch

F32_MIN = float(np.finfo(np.float32).min)
F32_MAX = float(np.finfo(np.float32).max)


# --------------------------------------------------------------------------- a1
def scorer(x: torch.Tensor, w1: torch.Tensor, b1: torch.Tensor,
           w2: torch.Tensor, b2: torch.Tensor) -> torch.Tensor:
    """scores[B,T] = w2 . relu(W1 x + b1) + b2   (w2v_asr.py:113-117,553).

    w1 is in nn.Linear layout [out, in]; w2 is [1, d]; b2 is [1].
    """
    h = torch.relu(x @ w1.t() + b1)
    return (h @ w2.t() + b2).squeeze(-1)


# --------------------------------------------------------------------------- a2
def nugget_counts(n_tok: np.ndarray, rate: float) -> np.ndarray:
    """k_b = trunc(fp32(n_tok) * fp32(1/rate) + 0.99), clamped to n_tok.

    w2v_asr.py:219-221.  The reference multiplies an int64 tensor by a Python
    float, which torch evaluates in fp32 with one rounding per op (mul, add).
    """
    n = np.asarray(n_tok, dtype=np.int64)
    ratio = np.float32(1.0 / rate)
    k = (n.astype(np.float32) * ratio).astype(np.float32) + np.float32(0.99)
    k = k.astype(np.float32).astype(np.int64)
    return np.minimum(k, n)


def topk_select(scores: np.ndarray, attn_mask: np.ndarray, rate: float | None = None,
                n_nugget: np.ndarray | None = None):
    """Per-utterance top-k anchor selection (w2v_asr.py:208-244).

    Returns (nugget_mask[B,K] bool, indices_ascending[B,K] int64,
    scores_gather[B,K] fp32).  Ordering rule restated: rank all T frames by
    (selection key descending, frame index ascending); keys are the scores with
    pad frames at finfo.min and the last valid frame forced to finfo.max.  The
    first k_b ranks are the kept set, ranks k_b..K-1 are the pad slots; each of
    the two groups is emitted in ascending frame order, kept group first.

    Ties: the reference ranks with torch's NON-stable CPU sort (w2v_asr.py:238), whose order
    among equal keys is an artefact of the sort kernel (e.g. 40 equal keys come back as
    [30, 21, 22, ...]); it only matters for pad slots that reach into padded frames and for
    exactly equal scores.  This restatement (and the HIP kernel) define the tie rule as
    "lower frame index first"; tests compare with the reference up to that tie rule.
    """
    scores = np.asarray(scores, dtype=np.float32)
    mask = np.asarray(attn_mask).astype(bool)
    bsz, t = scores.shape
    n_tok = mask.sum(axis=1).astype(np.int64)
    if n_nugget is None:
        n_nugget = nugget_counts(n_tok, rate)
    n_nugget = np.asarray(n_nugget, dtype=np.int64)
    kmax = int(n_nugget.max())
    keys = scores.copy()
    keys[~mask] = F32_MIN
    # python-style negative index when n_tok == 0 (torch scatter would raise); n_tok>=1 is required
    keys[np.arange(bsz), n_tok - 1] = F32_MAX
    idx_out = np.zeros((bsz, kmax), dtype=np.int64)
    frame = np.arange(t, dtype=np.int64)
    for b in range(bsz):
        # lexsort: last key is primary.  -keys turns "descending" into ascending and
        # maps -0.0/+0.0 to values that still compare equal, so ties fall to the frame index.
        order = np.lexsort((frame, -keys[b].astype(np.float64)))[:kmax]
        kb = int(n_nugget[b])
        idx_out[b, :kb] = np.sort(order[:kb])
        idx_out[b, kb:] = np.sort(order[kb:])
    nugget_mask = np.arange(kmax)[None, :] < n_nugget[:, None]
    scores_gather = np.take_along_axis(scores, idx_out, axis=1)
    return nugget_mask, idx_out, scores_gather


# --------------------------------------------------------------------------- a3
def type_table(t: int, nugget_indices: np.ndarray, nugget_mask: np.ndarray):
    """w2v_asr.py:316-323: 0/1 table [B,T] with ones at kept frames; indices with pads = -1."""
    idx = np.asarray(nugget_indices, dtype=np.int64).copy()
    idx[~np.asarray(nugget_mask, dtype=bool)] = -1
    bsz = idx.shape[0]
    table = np.zeros((bsz, t + 1), dtype=np.int64)
    for b in range(bsz):
        table[b, idx[b]] = 1          # -1 lands in the spare column t, sliced off below
    return table[:, :t], idx


# --------------------------------------------------------------------------- a4
def gather_rows(src: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """mem[b,j,:] = src[b, idx[b,j], :] with idx == -1 addressing the LAST frame
    (w2v_asr.py:566-567; w2v_simul.py:427,446-447).  Differentiable (scatter-add)."""
    b = torch.arange(src.shape[0]).unsqueeze(1)
    return src[b, idx]


def gather_layernorm(hidden: torch.Tensor, idx: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor,
                     eps: float = 1e-5) -> torch.Tensor:
    """The semantic encoder's final LayerNorm over every frame (my_transformer.py:945) followed by the nugget row
    gather (w2v_asr.py:566-567): ln_f(hidden)[arange(B)[:,None], idx], in the reference's order (normalise all T
    frames, then index).  Differentiable."""
    y = torch.nn.functional.layer_norm(hidden, (hidden.shape[-1],), weight, bias, eps)
    return gather_rows(y, idx)


# --------------------------------------------------------------------------- a5
def quantity_loss(scores: torch.Tensor, attn_mask: torch.Tensor,
                  target_lengths: torch.Tensor, threshold: float = 1.0):
    """alphas = sigmoid(s)*mask; q = mean_b((sum_t alphas/thr - L_b)^2)  (w2v_simul.py:419-424)."""
    alphas = torch.sigmoid(scores) * attn_mask
    q = ((alphas / threshold).sum(dim=1) - target_lengths) ** 2
    return alphas, q.mean()


# --------------------------------------------------------------------------- a6
def resize(alphas: torch.Tensor, target_lengths: torch.Tensor, threshold: float = 1.0,
           max_rounds: int = 10, differentiable: bool = False) -> torch.Tensor:
    """CIF weight rescale + halve-and-add-mean fix-ups (w2v_simul.py:504-535, noise = 0).

    Rows are independent, so the reference's batch-global loop is restated per row:
    each of up to ``max_rounds`` rounds snapshots the columns with value > thr, visits
    them in ascending order and, if the *current* value is still >= thr, replaces the row
    by 0.5*row + (0.5*sum(row)/nnz(row))*(row != 0).  A row with no offender in a round is
    finished (nothing else can change it).  Sums use torch's fp32 CPU reduction so that
    the restatement matches the reference bit-for-bit on the golden vectors.

    differentiable=True keeps the autograd graph, as the reference's own resize does (lm_cif_training
    back-propagates the LM loss through it, w2v_simul.py:558): the fix-up decisions and the mask row.ne(0) are
    constants of the graph, the scale and the row sums are not.
    """
    a = alphas.to(torch.float32) if differentiable else alphas.detach().to(torch.float32).clone()
    num = target_lengths.to(torch.float32)
    den = a.sum(-1)
    a = a * (num / den)[:, None]
    thr = float(threshold)
    rows = []
    for b in range(a.shape[0]):
        row = a[b]
        for _ in range(max_rounds):
            offenders = torch.nonzero(row.detach() > thr).flatten().tolist()
            if not offenders:
                break
            for y in offenders:
                if float(row.detach()[y]) >= thr:
                    nz = row.detach().ne(0).to(torch.float32)
                    mean = 0.5 * row.sum() / nz.sum()
                    row = row * 0.5 + mean * nz
        rows.append(row)
    return torch.stack(rows, 0)


def valid_mask_indices(mask: np.ndarray, padding_value: int = -1):
    """get_valid_mask_indices (w2v_simul.py:196-220): ascending positions of the True entries of each row, padded
    with padding_value to the largest count.  Returns (indices [B,Kmax] (or [Kmax] for a 1-D mask), Kmax, counts)."""
    m = np.asarray(mask).astype(bool)
    one_d = m.ndim == 1
    m2 = m[None, :] if one_d else m
    counts = m2.sum(axis=1).astype(np.int64)
    kmax = int(counts.max())
    idx = np.full((m2.shape[0], kmax), padding_value, dtype=np.int64)
    for b in range(m2.shape[0]):
        pos = np.flatnonzero(m2[b])
        idx[b, :pos.size] = pos
    return (idx[0], kmax, counts[0]) if one_d else (idx, kmax, counts)


# --------------------------------------------------------------------------- a7
def fire_scan(alphas: np.ndarray, attn_mask: np.ndarray, with_block_mask: bool = True):
    """Cumulative-score integrate-and-fire (w2v_simul.py:376-403 + 196-220).

    Sequential fp32 recurrence per row over ALL T frames (pads contribute 0 but can
    still fire if the carry is >= 1):  cur = fl(cur + a_t); fire = cur >= 1 or t == n_b-1;
    cur = fl(cur - 1) where fired.  Returns (nugget_mask[B,Kf] bool,
    nugget_indices[B,Kf] int64 ascending, -1 padded, block_attn_mask[B,T,T] fp32 or None)
    with block[b,i,j] = 1 iff  last_fire_before(i) < j <= i  and frame i is valid.
    """
    m = np.asarray(attn_mask).astype(bool)
    a = np.asarray(alphas, dtype=np.float32) * m.astype(np.float32)
    bsz, t = a.shape
    src_len = m.sum(axis=1).astype(np.int64)
    cur = np.zeros(bsz, dtype=np.float32)
    one = np.float32(1.0)
    fires = np.zeros((bsz, t), dtype=bool)
    for i in range(t):
        cur = (cur + a[:, i]).astype(np.float32)
        f = (cur >= one) | (src_len == i + 1)
        cur = np.where(f, (cur - one).astype(np.float32), cur)
        fires[:, i] = f
    counts = fires.sum(axis=1)
    kf = int(counts.max())
    idx = np.full((bsz, kf), -1, dtype=np.int64)
    for b in range(bsz):
        pos = np.flatnonzero(fires[b])
        idx[b, :pos.size] = pos
    nugget_mask = idx != -1
    block = None
    if with_block_mask:
        block = block_mask_from_fires(idx, m)
    return nugget_mask, idx, block


def block_mask_from_fires(nugget_indices: np.ndarray, attn_mask: np.ndarray) -> np.ndarray:
    """block[b,i,j] = 1.0 iff prev_fire(i) < j <= i and attn_mask[b,i]; prev_fire(i) is the
    largest fire position strictly below i (-1 if none).  w2v_simul.py:382-402."""
    m = np.asarray(attn_mask).astype(bool)
    bsz, t = m.shape
    out = np.zeros((bsz, t, t), dtype=np.float32)
    j = np.arange(t)
    for b in range(bsz):
        prev = -1
        fired = set(int(v) for v in nugget_indices[b] if v >= 0)
        for i in range(t):
            if m[b, i]:
                out[b, i] = ((j > prev) & (j <= i)).astype(np.float32)
            if i in fired:
                prev = i
    return out


# --------------------------------------------------------------------------- a8
def _mix32(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.uint32)
    x ^= x >> np.uint32(16)
    x = (x * np.uint32(0x7FEB352D)).astype(np.uint32)
    x ^= x >> np.uint32(15)
    x = (x * np.uint32(0x846CA68B)).astype(np.uint32)
    x ^= x >> np.uint32(16)
    return x


def dropout_keep_mask(seed: int, offset: int, bsz: int, heads: int, lq: int, lk: int, p: float):
    """The counter-based keep mask of the attention kernels, restated from its specification in
    include/star_nugget.h (star_xattn_shape).  NOT a reference algorithm: the reference draws its mask from
    torch's generator (nn.Dropout, my_transformer.py:595), which no other implementation can reproduce; what IS
    pinned against the reference is where the mask enters (xattn_core(drop_mask=...), golden xattn_drop*).
    Returns (keep [B,h,L,K] bool, p_eff)."""
    with np.errstate(over="ignore"):
        t = float(np.float32(p)) * 4294967296.0
        thr = np.uint32(0xFFFFFFFF) if t >= 4294967295.0 else np.uint32(int(t))
        p_eff = float(thr) / 4294967296.0
        seed, offset = int(seed) & (2 ** 64 - 1), int(offset) & (2 ** 64 - 1)
        bh = np.arange(bsz * heads, dtype=np.uint32)
        stream = (np.uint32(0x9E3779B9) * (bh + np.uint32(1))).astype(np.uint32)
        k1 = _mix32((np.uint32(offset >> 32) + stream).astype(np.uint32))
        k1 = _mix32(np.uint32(offset & 0xFFFFFFFF) ^ k1)
        k1 = _mix32(np.uint32(seed >> 32) ^ k1)
        k1 = _mix32(np.uint32(seed & 0xFFFFFFFF) ^ k1)
        k2 = _mix32(k1 ^ np.uint32(0x85EBCA6B))
        ctr = (np.arange(lq, dtype=np.uint32)[:, None] * np.uint32(lk) + np.arange(lk, dtype=np.uint32)[None, :])
        x = (k1[:, None, None] + ctr[None]).astype(np.uint32)
        x ^= x >> np.uint32(16)
        x = (x * np.uint32(0x7FEB352D)).astype(np.uint32)
        x ^= k2[:, None, None]
        x ^= x >> np.uint32(15)
        x = (x * np.uint32(0x846CA68B)).astype(np.uint32)
        x ^= x >> np.uint32(16)
    return (x >= thr).reshape(bsz, heads, lq, lk), p_eff


def xattn_core(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor,
               key_mask: torch.Tensor | None, scorer_logits: torch.Tensor | None,
               limit_cross_attn: bool = False, scale: bool = True,
               drop_mask: torch.Tensor | None = None, drop_p: float = 0.0,
               head_mask: torch.Tensor | None = None, inverse_layer_idx: int | None = None):
    """Score-biased cross-attention core (my_transformer.py:535-603, cross-attention branch).  head_mask (None on
    the reference's own call path): multiplied into the weights after the dropout (my_transformer.py:598-599).
    scale: scale_attn_weights (1/sqrt(hd), :538-541); inverse_layer_idx: layer_idx when
    scale_attn_by_inverse_layer_idx is set (a further 1/(layer_idx+1)), else None.

    q [B,h,L,hd], k/v [B,h,K,hd], key_mask [B,K] bool (True = attend), scorer_logits [B,Kb]
    with Kb <= K (zero-extended, my_transformer.py:114-125 variant).  drop_mask [B,h,L,K] (1 = keep) with
    drop_p: the attention-probability dropout of my_transformer.py:595 with the mask GIVEN, P <- P*mask/(1-p)
    after the softmax (and after the cast to V's dtype), before P.V.  Returns (O, P) with P the dropped weights,
    as the reference returns them.
    """
    a = q @ k.transpose(-1, -2)
    if scale:
        a = a / (v.shape[-1] ** 0.5)
    if inverse_layer_idx is not None:            # scale_attn_by_inverse_layer_idx (my_transformer.py:544-545)
        a = a / float(inverse_layer_idx + 1)
    lq, lk = a.shape[-2], a.shape[-1]
    fmin = torch.finfo(a.dtype).min
    if limit_cross_attn:
        i = torch.arange(lq).unsqueeze(1)
        j = torch.arange(lk).unsqueeze(0)
        allowed = (j - i) <= 2
        ilk = torch.where(allowed, torch.ones((), dtype=a.dtype), torch.full((), fmin, dtype=a.dtype))
        a = a + ilk
    if key_mask is not None:
        add = (1.0 - key_mask.to(a.dtype)) * fmin          # invert_attention_mask semantics
        a = a + add[:, None, None, :]
    if scorer_logits is not None:
        s = scorer_logits
        if s.shape[1] < lk:
            s = torch.cat([s, s.new_zeros(s.shape[0], lk - s.shape[1])], dim=1)
        a = a + s[:, None, None, :].to(a.dtype)
    p = torch.softmax(a, dim=-1)
    if drop_mask is not None:
        p = p * drop_mask.to(p.dtype) / (1.0 - drop_p)
    if head_mask is not None:
        p = p * head_mask
    return p @ v, p


# ------------------------------------------------------------------ 8f-3: nugget type embedding
def type_embed_add(x: torch.Tensor, table: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
    """inputs_embed = encoder_out.last_hidden_state + semantic_encoder.type_embed(type_ind_table)
    (w2v_asr.py:383,400,456,560; w2v_simul.py:432), type_embed = nn.Embedding(2, d) (my_transformer.py:761)."""
    return x + weight[table.long()]


# ------------------------------------------------------------------ 8f-2: CIF weighted aggregation
def cif_aggregate(alphas: torch.Tensor, s: torch.Tensor, threshold: float = 1.0):
    """Continuous integrate-and-fire weighted aggregation (w2v_simul.py:558-606; same loop at w2v_asr.py:681-722).

    Serial fp32 recurrence per utterance: dc = 1 - I; I += a_t; fire = I >= thr; I -= 1 where fired;
    cur = dc where fired else a_t; rem = a_t - cur; frame += cur * s_t; the frame is EMITTED at a fire and restarts
    as rem * s_t.  Unlike the nugget fire scan there is no forced fire at the last valid frame, and the partial
    segment after the last fire is dropped.  Differentiable w.r.t. alphas and s (fire decisions are constants).
    Returns (cif_repr [B,Kf,d], cif_mask [B,Kf] fp32 0/1, fired_positions [B,Kf] int64 ascending, -1 padded)."""
    bsz, t_len = alphas.shape
    integ = alphas.new_zeros(bsz)
    frame = s.new_zeros(bsz, s.shape[-1])
    frames, fired = [], []
    for t in range(t_len):
        a = alphas[:, t]
        dc = 1.0 - integ
        integ = integ + a
        fire = integ >= threshold
        integ = torch.where(fire, integ - 1.0, integ)
        cur = torch.where(fire, dc, a)
        rem = a - cur
        frame = frame + cur[:, None] * s[:, t, :]
        frames.append(frame)
        frame = torch.where(fire[:, None], rem[:, None] * s[:, t, :], frame)
        fired.append(fire)
    frames = torch.stack(frames, 1)
    fired = torch.stack(fired, 1)
    counts = fired.sum(1)
    kf = int(counts.max())
    rows, pos_all = [], torch.full((bsz, kf), -1, dtype=torch.int64)
    mask = alphas.new_zeros(bsz, kf)
    for b in range(bsz):
        pos = torch.where(fired[b])[0]
        sel = frames[b].index_select(0, pos)
        rows.append(torch.cat([sel, s.new_zeros(kf - pos.numel(), s.shape[-1])], 0))
        pos_all[b, :pos.numel()] = pos
        mask[b, :pos.numel()] = 1
    return torch.stack(rows, 0), mask, pos_all


# ------------------------------------------------------------------ composite step (bench/smoke)
def nugget_step_asr(x, s_repr, lengths, rate, w1, b1, w2, b2, qs, kv_scales, gs, select=None):
    """One NuggetStep of the non-streaming path, fp32 on CPU.

    x [B,T,d] scorer input, s_repr [B,T,d] semantic states, lengths list[int], qs list of
    [B,h,L,hd] queries (one per layer), kv_scales list of (ak[d], av[d]) fixed diagonal K/V
    maps, gs list of cotangents like qs.  Returns dict with scores, indices, loss.

    select = (nugget_mask, indices) replaces the oracle's own selection (a test passes the kernel's, which it has
    checked bit for bit against topk_select on the kernel's own scores, so that a near-tie resolved differently by the
    bf16 and the fp32 scorer cannot hide everything downstream); "own_indices" always holds the oracle's.
    """
    bsz, t, d = x.shape
    h = qs[0].shape[1]
    mask = torch.arange(t)[None, :] < torch.tensor(lengths)[:, None]
    scores = scorer(x, w1, b1, w2, b2)
    nmask, idx, _ = topk_select(scores.detach().numpy(), mask.numpy(), rate)
    own_idx = idx
    if select is not None:
        nmask, idx = (np.asarray(v) for v in select)
        nmask, idx = nmask.astype(bool), idx.astype(np.int64)
    idx_t = torch.from_numpy(idx)
    nmask_t = torch.from_numpy(nmask)
    sg = scores.gather(1, idx_t)
    table, pos = type_table(t, idx, nmask)
    mem = gather_rows(s_repr, idx_t)
    loss = x.new_zeros(())
    for q, (ak, av), g in zip(qs, kv_scales, gs):
        kk = (mem * ak).view(bsz, -1, h, d // h).transpose(1, 2)
        vv = (mem * av).view(bsz, -1, h, d // h).transpose(1, 2)
        o, _ = xattn_core(q, kk, vv, nmask_t, sg, False)
        loss = loss + (o * g).sum()
    return {"scores": scores, "indices": idx, "own_indices": own_idx, "mask": nmask, "table": table, "positions": pos,
            "scores_gather": sg, "mem": mem, "loss": loss}


def nugget_step_simul(x, s_repr, lengths, target_lengths, use_ilk, w1, b1, w2, b2, qs, kv_scales, gs,
                      fires=None):
    """One NuggetStep of the simultaneous path (w2v_simul.py:407-461), fp32 on CPU: scorer on the detached encoder
    states (:410-411), alphas and quantity loss (:419-423), resize (:425), fire scan on the detached resized alphas
    (:426), score gather (:427), type table (:431), row gather (:446-447), nl x score-biased cross-attention (ILK mask
    j <= i + 2 when use_ilk), loss = sum_l <O_l, G_l> + 0.01 * quantity (:461).

    fires = (nugget_mask, nugget_indices) replaces the oracle's own fire positions (a test passes the kernel's, which
    it has checked separately, so that a one-ulp score difference cannot move a fire and hide everything after it)."""
    bsz, t, d = x.shape
    h = qs[0].shape[1]
    mask = torch.arange(t)[None, :] < torch.tensor(lengths)[:, None]
    tl = torch.as_tensor(target_lengths)
    scores = scorer(x.detach(), w1, b1, w2, b2)
    alphas, qloss = quantity_loss(scores, mask, tl)
    resized = resize(alphas, tl)
    if fires is None:
        nmask, idx, _ = fire_scan(resized.numpy(), mask.numpy(), with_block_mask=False)
    else:
        nmask, idx = (np.asarray(v) for v in fires)
    idx_t = torch.from_numpy(idx.astype(np.int64))
    nmask_t = torch.from_numpy(nmask.astype(bool))
    sg = gather_rows(scores.unsqueeze(-1), idx_t).squeeze(-1)
    table, pos = type_table(t, idx, nmask)
    mem = gather_rows(s_repr, idx_t)
    loss = x.new_zeros(())
    for q, (ak, av), g in zip(qs, kv_scales, gs):
        kk = (mem * ak).view(bsz, -1, h, d // h).transpose(1, 2)
        vv = (mem * av).view(bsz, -1, h, d // h).transpose(1, 2)
        o, _ = xattn_core(q, kk, vv, nmask_t, sg, bool(use_ilk))
        loss = loss + (o * g).sum()
    return {"scores": scores, "alphas": alphas, "resized": resized, "quantity_loss": qloss, "indices": idx,
            "mask": nmask, "table": table, "positions": pos, "scores_gather": sg, "mem": mem,
            "loss": loss + 0.01 * qloss}


# ------------------------------------------------------------------ configs[0]: the whole-decoder CPU step (timing only)
class DecoderStepConfig0:
    """CPU timing stand-in for BASELINE.json configs[0] as the survey measured it on the reference (SURVEY.md section
    6 and 8d: scorer, select, gather, then the WHOLE lm_decoder forward + backward, 124.3 ms on 8 threads): the
    NuggetStep pieces above inside a GPT-2-shaped decoder with the reference's configuration (w2v_asr.py:134-145:
    n_head 8, n_positions 512, add_cross_attention; BASELINE configs[0]: 6 layers, vocabulary 10 000) and block
    structure (my_transformer.py:686-750: ln_1, causal self-attention, ln_cross_attn, score-biased cross-attention,
    ln_2, MLP; :945 ln_f; :1041-1058 classifier + shifted cross-entropy).  Used ONLY by bench.py's cpu_baseline leg
    for `--config 0`; no parity claim rests on it (the decoder scaffolding is outside the hot path; its cross-attention
    core is xattn_core above, which is pinned).  Calling the object runs one forward + backward."""

    def __init__(self, lengths, t, d, h, l_tgt, nl, rate, vocab=10000, seed=0):
        g = torch.Generator().manual_seed(seed)
        self.lengths, self.rate, self.h, self.nl, self.d = list(lengths), rate, h, nl, d
        bsz = len(lengths)
        self.x = torch.randn(bsz, t, d, generator=g).requires_grad_()
        self.s = torch.randn(bsz, t, d, generator=g).requires_grad_()
        self.ids = torch.randint(0, vocab, (bsz, l_tgt), generator=g)

        def p(*shape, std=0.02):
            return (torch.randn(*shape, generator=g) * std).requires_grad_()
        lim = 1 / d ** 0.5
        self.scorer = [((torch.rand(d, d, generator=g) * 2 - 1) * lim).requires_grad_(),
                       ((torch.rand(d, generator=g) * 2 - 1) * lim).requires_grad_(),
                       ((torch.rand(1, d, generator=g) * 2 - 1) * lim).requires_grad_(),
                       ((torch.rand(1, generator=g) * 2 - 1) * lim).requires_grad_()]
        self.wte, self.wpe = p(vocab, d), p(512, d)
        self.layers = []
        for _ in range(nl):
            self.layers.append(dict(
                ln1=(torch.ones(d).requires_grad_(), torch.zeros(d).requires_grad_()),
                attn_w=p(d, 3 * d), attn_b=torch.zeros(3 * d).requires_grad_(),
                attn_pw=p(d, d), attn_pb=torch.zeros(d).requires_grad_(),
                lnx=(torch.ones(d).requires_grad_(), torch.zeros(d).requires_grad_()),
                q_w=p(d, d), q_b=torch.zeros(d).requires_grad_(),
                kv_w=p(d, 2 * d), kv_b=torch.zeros(2 * d).requires_grad_(),
                x_pw=p(d, d), x_pb=torch.zeros(d).requires_grad_(),
                ln2=(torch.ones(d).requires_grad_(), torch.zeros(d).requires_grad_()),
                fc_w=p(d, 4 * d), fc_b=torch.zeros(4 * d).requires_grad_(),
                fc2_w=p(4 * d, d), fc2_b=torch.zeros(d).requires_grad_()))
        self.lnf = (torch.ones(d).requires_grad_(), torch.zeros(d).requires_grad_())
        self.cls_w, self.cls_b = p(d, vocab), torch.zeros(vocab).requires_grad_()

    def _leaves(self):
        out = [self.x, self.s, self.wte, self.wpe, self.cls_w, self.cls_b, *self.scorer, *self.lnf]
        for lay in self.layers:
            for v in lay.values():
                out.extend(v if isinstance(v, tuple) else [v])
        return out

    def __call__(self):
        ln = torch.nn.functional.layer_norm
        for t in self._leaves():
            t.grad = None
        bsz, t, d = self.x.shape
        h, hd = self.h, d // self.h
        mask = torch.arange(t)[None, :] < torch.tensor(self.lengths)[:, None]
        scores = scorer(self.x, *self.scorer)
        nmask, idx, _ = topk_select(scores.detach().numpy(), mask.numpy(), self.rate)
        idx_t, nmask_t = torch.from_numpy(idx), torch.from_numpy(nmask)
        sg = scores.gather(1, idx_t)
        mem = gather_rows(self.s, idx_t)
        lt = self.ids.shape[1]
        hid = self.wte[self.ids] + self.wpe[:lt][None]
        causal = torch.tril(torch.ones(lt, lt, dtype=torch.bool))

        def heads(u):
            return u.view(bsz, -1, h, hd).transpose(1, 2)
        for lay in self.layers:
            a = ln(hid, (d,), *lay["ln1"])
            q, k, v = (a @ lay["attn_w"] + lay["attn_b"]).split(d, dim=2)
            att = heads(q) @ heads(k).transpose(-1, -2) / hd ** 0.5
            att = torch.softmax(att.masked_fill(~causal, F32_MIN), dim=-1)
            hid = hid + ((att @ heads(v)).transpose(1, 2).reshape(bsz, lt, d) @ lay["attn_pw"] + lay["attn_pb"])
            a = ln(hid, (d,), *lay["lnx"])
            q = a @ lay["q_w"] + lay["q_b"]
            k, v = (mem @ lay["kv_w"] + lay["kv_b"]).split(d, dim=2)
            o, _ = xattn_core(heads(q), heads(k), heads(v), nmask_t, sg, False)
            hid = hid + (o.transpose(1, 2).reshape(bsz, lt, d) @ lay["x_pw"] + lay["x_pb"])
            a = ln(hid, (d,), *lay["ln2"])
            hid = hid + (torch.nn.functional.gelu(a @ lay["fc_w"] + lay["fc_b"], approximate="tanh") @ lay["fc2_w"]
                         + lay["fc2_b"])
        logits = ln(hid, (d,), *self.lnf) @ self.cls_w + self.cls_b
        loss = torch.nn.functional.cross_entropy(logits[:, :-1].reshape(-1, logits.shape[-1]), self.ids[:, 1:].reshape(-1))
        loss.backward()
        return loss
