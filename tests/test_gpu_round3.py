"""GPU parity, round-3 additions (all through the C ABI): the exact-1.0 corner of the two-instruction fire chain,
whole non-streaming steps at configs[1]'s other compress rates (2 / 8 / 12) at full size, configs[0] with its six
decoder layers.

Bars as in tests/test_gpu_parity.py: integer results bit-exact; bf16-MFMA results against the fp32 oracle on the same
bf16-rounded inputs at 2e-2..3e-2 of the tensor scale."""
import numpy as np
import pytest
import torch

from oracle import nugget_oracle as orc
from tests.test_oracle_golden import lmask

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    from star_amd import ops as o
    return o


def bf16_round(t):
    return t.to(torch.bfloat16).to(torch.float32)


# ------------------------------------------------------------------------------------------------ fire chain corner
ALMOST_ONE = np.array([0x3F7FFFFF], dtype=np.uint32).view(np.float32)[0]        # 1 - 2^-24


def test_fire_step_weight_one_on_a_carry_just_below_one(ops):
    """w2v_simul.py:386-389 with cur = 1 - 2^-24 and a weight of exactly 1.0: the fp32 sum rounds (ties to even) to
    2.0, the reference subtracts 1 and keeps a carry of 1.0, so the NEXT frame fires too whatever its weight.  A
    v_fract chain would return fract(2.0) = 0 and lose that fire: rows that hold an exact 1.0 take the generic chain."""
    chunk = np.array([[1.0, 0.5, 0.0], [1.0, 0.0, 0.0], [0.5, 0.5, 0.0], [1.0, 1.0, 1.0]], dtype=np.float32)
    B, Tc = chunk.shape
    # the recurrence itself, in numpy fp32 (no forced fire: the end of the utterance is unknown)
    want, carry = [], []
    for b in range(B):
        cur, f = ALMOST_ONE, []
        for t in range(Tc):
            cur = np.float32(cur + chunk[b, t])
            if cur >= 1:
                f.append(t)
                cur = np.float32(cur - np.float32(1))
        want.append(f)
        carry.append(cur)
    assert want[0] == [0, 1] and want[1] == [0, 1] and want[2] == [0, 1] and want[3] == [0, 1, 2]
    st = ops.FireState(B, DEV)
    st.cur.copy_(torch.from_numpy(np.full(B, ALMOST_ONE, dtype=np.float32)))
    unknown = torch.full((B,), 2 ** 31 - 1, dtype=torch.int32, device=DEV)
    idx, cnt, _ = ops.cif_fire_step(torch.from_numpy(chunk).to(DEV), unknown, st, cap=Tc)
    idx, cnt = idx.cpu().numpy(), cnt.cpu().numpy()
    for b in range(B):
        assert [int(v) for v in idx[b, :cnt[b]]] == want[b], (b, idx[b], want[b])
    np.testing.assert_array_equal(st.cur.cpu().numpy(), np.asarray(carry, dtype=np.float32))
    # the oracle's scan agrees on the same recurrence (carry folded into a leading frame)
    full = np.concatenate([np.full((B, 1), ALMOST_ONE, dtype=np.float32), chunk], axis=1)
    o_mask, o_idx, _ = orc.fire_scan(full, np.ones_like(full, dtype=bool), with_block_mask=False)
    # whole-utterance scan through the kernel, same rows (the last frame is a forced fire in both)
    widx, _, wcnt, _, _ = ops.cif_fire_scan(torch.from_numpy(full).to(DEV),
                                            lengths=torch.full((B,), Tc + 1, dtype=torch.int32, device=DEV), cap=Tc + 1)
    widx, wcnt = widx.cpu().numpy(), wcnt.cpu().numpy()
    for b in range(B):
        np.testing.assert_array_equal(widx[b, :wcnt[b]], o_idx[b, :o_mask[b].sum()])


def test_cif_segment_row_with_an_exact_one(ops):
    """The same corner through star_cif_segment: scores whose sigmoids are [.5, .5 - 2^-24, 1, .5, .5, 0, .5, .5] sum
    to 4 after rounding, so resize scales by exactly 1 and the scan meets cur = 1 - 2^-24 followed by a weight of 1.0:
    fires at frames 2, 3, 4 and 7 (a fract chain gives 2, 4, 7)."""
    # find the score whose device sigmoid is exactly 0.5 - 2^-24 (expf rounding decides; -2^-22 on paper)
    cand = torch.tensor([[-k * 2.0 ** -24 for k in range(1, 33)]], dtype=torch.float32, device=DEV)
    al, _ = ops.cif_quantity(cand, torch.ones(1, dtype=torch.int32, device=DEV),
                             lengths=torch.tensor([cand.shape[1]], dtype=torch.int32, device=DEV))
    want_a = np.float32(0.5) - np.float32(2.0 ** -24)
    hit = (al[0].cpu().numpy() == want_a).nonzero()[0]
    assert hit.size > 0, "no score gives sigmoid = 0.5 - 2^-24 on this device"
    s_eps = float(cand[0, hit[0]])
    scores = np.array([[0.0, s_eps, 40.0, 0.0, 0.0, -200.0, 0.0, 0.0],
                       [0.0, 0.0, 40.0, 0.0, 0.0, -200.0, 0.0, 0.0]], dtype=np.float32)   # second row: an ordinary carry
    B, T = scores.shape
    ln = torch.tensor([T, T], dtype=torch.int32, device=DEV)
    tl = torch.tensor([4, 4], dtype=torch.int32, device=DEV)
    alphas, qloss, resized, idx, nmask, counts, mx, sg, table, pos = ops.cif_segment(
        torch.from_numpy(scores).to(DEV), tl, ln, cap=T)
    a = alphas.cpu().numpy()
    np.testing.assert_array_equal(a[0], np.array([.5, want_a, 1, .5, .5, 0, .5, .5], dtype=np.float32))
    r = resized.cpu().numpy()
    np.testing.assert_array_equal(r[0], a[0])                       # scale = 4 / 4 exactly, no fix-up (1.0 is not > 1)
    o_mask, o_idx, _ = orc.fire_scan(r, np.ones((B, T), dtype=bool), with_block_mask=False)
    idx, counts = idx.cpu().numpy(), counts.cpu().numpy()
    assert [int(v) for v in idx[0, :counts[0]]] == [2, 3, 4, 7]
    for b in range(B):
        np.testing.assert_array_equal(idx[b, :counts[b]], o_idx[b, :o_mask[b].sum()])
    # and through the stand-alone scan
    widx, _, wcnt, _, _ = ops.cif_fire_scan(resized, lengths=ln, cap=T)
    assert torch.equal(widx.cpu()[:, :idx.shape[1]], torch.from_numpy(idx)) and torch.equal(wcnt.cpu(), torch.from_numpy(counts))


# ------------------------------------------------------------------------------------------------ configs[1], rates
@pytest.mark.parametrize("rate", [2, 8, 12])
def test_full_size_asr_step_at_rate(ops, rate):
    """BASELINE.json configs[1] names compress rates 2 / 4 / 8 / 12 (attention K = 750 / 375 / 188 / 125, scorer
    backward over 24 k / 12 k / 6 k / 4 k frames).  The whole non-streaming step at full size (B32 x T1500 x d1024, 8
    heads, L128, 4 layers), ragged lengths: selection bit-exact against the oracle on the kernel's own scores; exact
    copies; gradients land where the reference's do; the whole backward is exactly linear in the cotangent; and one
    utterance of the batch against the fp32 oracle step on the same bf16-rounded inputs."""
    from star_amd.nugget import NuggetScorer
    from star_amd.step import make_inputs, nugget_step_asr
    torch.manual_seed(20 + rate)
    B, T, d, h, L, nl = 32, 1500, 1024, 8, 128, 4
    rng = np.random.default_rng(rate)
    lengths = [T] + [int(v) for v in rng.integers(T // 3, T + 1, size=B - 1)]
    scorer = NuggetScorer(d).to(DEV)
    inp = make_inputs(lengths, T, d, h, L, nl, DEV, seed=77 + rate)
    out = nugget_step_asr(scorer, inp, rate, with_loss=True)
    kb = orc.nugget_counts(np.asarray(lengths), rate)
    K = int(kb.max())
    assert K == {2: 750, 8: 188, 12: 125}[rate] and out["indices"].shape == (B, K)
    mask = lmask(lengths, T)
    scores = out["scores"].detach().cpu().numpy()
    assert (scores[~mask] == 0).all()
    # a2 / a3: bit-exact against the oracle on the scores of the same scorer implementation
    o_mask, o_idx, o_sg = orc.topk_select(scores, mask, rate)
    idx = out["indices"].cpu().numpy()
    np.testing.assert_array_equal(idx, o_idx)
    np.testing.assert_array_equal(out["nugget_mask"].cpu().numpy(), o_mask)
    np.testing.assert_array_equal(out["scores_gather"].detach().cpu().numpy(), o_sg)
    tab = out["type_table"].cpu().numpy()
    np.testing.assert_array_equal(tab.sum(1), kb)
    pos = out["positions"].cpu().numpy()
    assert (pos[o_mask] == idx[o_mask]).all() and (pos[~o_mask] == -1).all()
    kept = np.zeros((B, T), dtype=bool)
    for b in range(B):
        kept[b, idx[b, :kb[b]]] = True
    np.testing.assert_array_equal(tab.astype(bool), kept)
    # a4: rows are copies
    for b, j in zip(rng.integers(0, B, size=32), rng.integers(0, K, size=32)):
        assert torch.equal(out["mem"][b, j], inp.s[b, idx[b, j]])
    # backward: S and X receive gradient exactly at the kept frames (pad slots carry P = 0 -> exact zeros)
    ds = inp.s.grad.float().abs().sum(-1).cpu().numpy()
    assert (ds[~kept] == 0).all() and (ds[kept] > 0).mean() > 0.99
    dx = inp.x.grad.float().abs().sum(-1).cpu().numpy()
    assert (dx[~kept] == 0).all() and (dx[kept] > 0).mean() > 0.99
    grads1 = [p.grad.clone() for p in scorer.parameters()] + [inp.x.grad.clone(), inp.s.grad.clone()] + \
             [q.grad.clone() for q in inp.qs]
    for g in grads1:
        assert torch.isfinite(g).all() and g.abs().max() > 0
    # exact linearity: doubling every cotangent doubles every gradient bit for bit (a power of two commutes with
    # every rounding of the backward: bf16 dS / G', fp32 accumulators)
    leaves = list(scorer.parameters()) + [inp.x, inp.s] + inp.qs
    for t in leaves:
        t.grad = None
    saved = [g.clone() for g in inp.gs]
    for g in inp.gs:
        g.mul_(2.0)
    nugget_step_asr(scorer, inp, rate)
    for g, s_ in zip(inp.gs, saved):
        g.copy_(s_)
    for t, g1 in zip(leaves, grads1):
        assert torch.equal(t.grad, g1 * 2), "backward is not exactly linear in the cotangent"
    # one utterance against the fp32 oracle step (utterances are independent: its activations' gradients do not
    # depend on the rest of the batch)
    b = 3
    cpu = lambda t: t.detach().float().cpu()
    w1r = bf16_round(cpu(scorer[0].weight)).requires_grad_()
    b1, w2, b2 = (cpu(p).requires_grad_() for p in (scorer[0].bias, scorer[2].weight, scorer[2].bias))
    xr, sr = cpu(inp.x[b:b + 1]).requires_grad_(), cpu(inp.s[b:b + 1]).requires_grad_()
    qs = [cpu(q[b:b + 1]).requires_grad_() for q in inp.qs]
    kk = kb[b]
    ref = orc.nugget_step_asr(xr, sr, [lengths[b]], rate, w1r, b1, w2, b2, qs, [(cpu(a), cpu(c)) for a, c in inp.kv_scales],
                              [cpu(g[b:b + 1]) for g in inp.gs], select=(o_mask[b:b + 1, :kk], idx[b:b + 1, :kk]))
    nb = lengths[b]
    np.testing.assert_allclose(scores[b, :nb], ref["scores"].detach().numpy()[0, :nb], rtol=2e-3, atol=2e-3)
    # the fp32 scorer's own selection may resolve a near-tie differently; it must agree on all but a few frames
    diff = len(set(ref["own_indices"][0, :kk].tolist()) ^ set(idx[b, :kk].tolist())) // 2
    assert diff <= max(2, kk // 50), f"{diff} of {kk} selected frames differ from the fp32 oracle's selection"
    ref["loss"].backward()

    def close(a, r_, what, tol=3e-2):
        a, r_ = a.float().cpu().numpy(), r_.numpy()
        scale = np.abs(r_).max() + 1e-12
        assert np.abs(a - r_).max() <= tol * scale, f"{what}: {np.abs(a - r_).max()} vs scale {scale}"
    close(grads1[5][b:b + 1], sr.grad, "dS")
    close(grads1[4][b:b + 1], xr.grad, "dX")
    for l in range(nl):
        close(grads1[6 + l][b:b + 1], qs[l].grad, f"dQ{l}")


# ------------------------------------------------------------------------------------------------ configs[0]
def test_config0_six_layer_step_vs_oracle(ops):
    """BASELINE.json configs[0] as written: B = 4, T = 250 (lengths 250/230/200/170), d = 768 (head dim 96: the
    register-staged attention kernels), compress rate 4, SIX decoder layers; every gradient against the fp32 oracle."""
    from star_amd.nugget import NuggetScorer
    from star_amd.step import make_inputs, nugget_step_asr
    torch.manual_seed(0)
    lengths, T, d, h, L, nl, rate = [250, 230, 200, 170], 250, 768, 8, 24, 6, 4
    scorer = NuggetScorer(d).to(DEV)
    inp = make_inputs(lengths, T, d, h, L, nl, DEV)
    out = nugget_step_asr(scorer, inp, rate, with_loss=True)
    scores = out["scores"].detach().cpu().numpy()
    mask = lmask(lengths, T)
    o_mask, o_idx, _ = orc.topk_select(scores, mask, rate)
    np.testing.assert_array_equal(out["indices"].cpu().numpy(), o_idx)
    np.testing.assert_array_equal(out["nugget_mask"].cpu().numpy(), o_mask)
    assert o_idx.shape[1] == 63
    cpu = lambda t: t.detach().float().cpu()
    w1, b1, w2, b2 = (cpu(p).requires_grad_() for p in (scorer[0].weight, scorer[0].bias, scorer[2].weight, scorer[2].bias))
    w1r = bf16_round(w1.detach()).requires_grad_()
    xr, sr = cpu(inp.x).requires_grad_(), cpu(inp.s).requires_grad_()
    qs = [cpu(q).requires_grad_() for q in inp.qs]
    # the oracle runs on the kernel's selection (checked bit for bit above on the kernel's own scores): its own fp32
    # scores may resolve a near-tie differently, which must stay a handful of frames
    ref = orc.nugget_step_asr(xr, sr, lengths, rate, w1r, b1, w2, b2, qs, [(cpu(a), cpu(b)) for a, b in inp.kv_scales],
                              [cpu(g) for g in inp.gs], select=(o_mask, o_idx))
    np.testing.assert_allclose(scores[mask], ref["scores"].detach().numpy()[mask], rtol=2e-3, atol=2e-3)
    assert (ref["own_indices"] != o_idx).sum() <= 8
    ref["loss"].backward()
    assert abs(out["loss"].item() - ref["loss"].item()) <= 2e-2 * abs(ref["loss"].item()) + 1e-2

    def close(a, b, what, tol=3e-2):
        a, b = a.float().cpu().numpy(), b.numpy()
        scale = np.abs(b).max() + 1e-12
        assert np.abs(a - b).max() <= tol * scale, f"{what}: {np.abs(a - b).max()} vs scale {scale}"
    close(scorer[0].weight.grad, w1r.grad, "dW1")
    close(scorer[2].weight.grad, w2.grad, "dw2")
    close(scorer[0].bias.grad, b1.grad, "db1")
    # the select step's true db2 is 0 (the softmax is invariant under a common shift of the logits): one cancelling sum,
    # measured on the scale of its sibling dw2[n] = sum_r dscore[r] h[r][n]
    assert abs(scorer[2].bias.grad.item() - b2.grad.item()) <= 3e-2 * float(w2.grad.abs().max())
    close(inp.s.grad, sr.grad, "dS")
    close(inp.x.grad, xr.grad, "dX")
    for l in range(nl):
        close(inp.qs[l].grad, qs[l].grad, f"dQ{l}")


# ------------------------------------------------------------------------------------------------ f4: streaming
def _offline_memory(ops, scores, s, lengths):
    """Whole-utterance kernels: alphas = sigmoid * mask, fire scan with the score gather, row gather."""
    ln = torch.tensor(lengths, dtype=torch.int32, device=DEV)
    alphas, _ = ops.cif_quantity(scores, torch.zeros_like(ln), lengths=ln)
    idx, nmask, counts, _, sg = ops.cif_fire_scan(alphas, lengths=ln, cap=scores.shape[1], scores=scores)
    return idx, counts, sg, ops.gather_rows(s, idx)


def _check_memory(state, idx, counts, sg, mem_off, what=""):
    assert torch.equal(state.counts, counts), (what, state.counts.tolist(), counts.tolist())
    for b in range(idx.shape[0]):
        n = int(counts[b])
        assert torch.equal(state.mem_pos[b, :n], idx[b, :n]), (what, b)
        assert torch.equal(state.mem[b, :n], mem_off[b, :n]), (what, b)
        assert torch.equal(state.mem_scores[b, :n], sg[b, :n]), (what, b)
    assert int(state.dropped.sum()) == 0


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_stream_push_any_chunking_equals_offline(ops, dtype):
    """star_cif_stream_push (sigmoid * mask + fire recurrence + append in one launch) against the offline kernels:
    memory rows, logits, positions and counts bit for bit for any chunking, with the end of an utterance known from
    the start, with the chunk that holds its last frame, or one push late (the late flush inside the next push; after
    the last chunk the Tc = 0 flush); per-chunk fire lists equal star_cif_fire_step's."""
    rng = np.random.default_rng(11)
    B, T, d = 6, 700, 256 if dtype == torch.bfloat16 else 64
    g = torch.Generator().manual_seed(3)
    scores = (torch.randn(B, T, generator=g) * 1.5 - 2.0)
    scores[1] = 25.0                                  # sigmoid = 1.0 exactly: every frame fires (generic chain)
    scores[2, 100:140] = -200.0                       # exact zeros inside the utterance
    scores = scores.to(DEV)
    s = torch.randn(B, T, d, generator=g).to(dtype).to(DEV)
    lengths = [700, 512, 640, 64, 1, 333]
    idx, counts, sg, mem_off = _offline_memory(ops, scores, s, lengths)
    kmax = int(counts.max()) + 3
    ln = torch.tensor(lengths, dtype=torch.int32, device=DEV)
    unknown = torch.full_like(ln, 2 ** 31 - 1)
    chunkings = [[T], [64] * 10 + [60], [1] * 64 + [636], [333, 179, 128, 60]]
    for _ in range(2):
        sizes, left = [], T
        while left:
            c = int(min(left, rng.integers(1, 160)))
            sizes.append(c)
            left -= c
        chunkings.append(sizes)
    for chunks in chunkings:
        for mode in ("known", "with_last_frame", "late"):
            st = ops.StreamState(B, d, kmax, DEV, dtype)
            fs = ops.FireState(B, DEV)
            off = 0
            for c in chunks:
                if mode == "known":
                    cur = ln
                elif mode == "with_last_frame":
                    cur = torch.where(ln <= off + c, ln, unknown)
                else:
                    # an utterance that ends exactly with this chunk stays unknown: it is announced one push late (the
                    # flush inside the next push) or, at the end of the stream, by the Tc = 0 flush.  One that ends
                    # inside the chunk must be announced with it (its later frames would otherwise count as valid)
                    cur = torch.where(ln < off + c, ln, unknown)
                sc_c = scores[:, off:off + c].contiguous()
                io = torch.empty(B, c, dtype=torch.int64, device=DEV)
                co = torch.empty(B, dtype=torch.int32, device=DEV)
                ops.cif_stream_push(sc_c, s[:, off:off + c].contiguous(), cur.contiguous(), st, idx_out=io, cnt_out=co)
                if mode in ("known", "with_last_frame"):
                    rel = (cur.to(torch.int64) - off).clamp(0, c).to(torch.int32)
                    al, _ = ops.cif_quantity(sc_c, torch.zeros_like(ln), lengths=rel)
                    i2, c2, _ = ops.cif_fire_step(al, cur.contiguous(), fs, cap=c)
                    assert torch.equal(io, i2) and torch.equal(co, c2), (chunks[:4], mode, off)
                off += c
            if mode == "late":
                ops.cif_stream_push(None, None, ln, st)                       # the stream has stopped: flush
            assert (st.pos == T).all()
            _check_memory(st, idx, counts, sg, mem_off, (chunks[:4], mode))


def test_stream_push_overflow_is_counted(ops):
    """max_nuggets too small: the memory keeps the first kmax nuggets, counts stop at kmax and `dropped` holds the
    number of fires that found no slot (no silent truncation)."""
    B, T, d = 3, 400, 128
    g = torch.Generator().manual_seed(5)
    scores = (torch.randn(B, T, generator=g) - 1.0).to(DEV)
    s = torch.randn(B, T, d, generator=g).to(torch.bfloat16).to(DEV)
    lengths = [400, 250, 30]
    idx, counts, sg, mem_off = _offline_memory(ops, scores, s, lengths)
    kmax = 20
    assert int(counts.max()) > kmax > int(counts.min())
    st = ops.StreamState(B, d, kmax, DEV)
    ln = torch.tensor(lengths, dtype=torch.int32, device=DEV)
    for off in range(0, T, 50):
        ops.cif_stream_push(scores[:, off:off + 50].contiguous(), s[:, off:off + 50].contiguous(), ln, st)
    want = torch.clamp(counts, max=kmax)
    assert torch.equal(st.counts, want)
    assert torch.equal(st.dropped, counts - want)
    for b in range(B):
        n = int(want[b])
        assert torch.equal(st.mem_pos[b, :n], idx[b, :n]) and torch.equal(st.mem[b, :n], mem_off[b, :n])


def test_streaming_decode_token_by_token_matches_offline_ilk(ops):
    """SURVEY.md 8f row 4 end to end: chunks in -> StreamingNuggetSegmenter -> StreamingEncoderKV (incremental c_attn
    of all layers) -> NuggetCrossAttention one token at a time (encoder_kv = the cache, query_offset = the token's
    index) as soon as tokens_ready() allows it, against the offline module call on the whole memory with
    limit_cross_attn (my_transformer.py:563-570): every token's output row matches."""
    from star_amd.nugget import NuggetCrossAttention, NuggetScorer
    from star_amd.streaming import StreamingEncoderKV, StreamingNuggetSegmenter
    torch.manual_seed(4)
    B, T, d, h, nl, Ltok = 3, 960, 256, 4, 2, 40
    lengths = [960, 700, 200]
    scorer = NuggetScorer(d).to(DEV)
    with torch.no_grad():
        scorer[2].bias.fill_(-2.2)
    mods = [NuggetCrossAttention(d, h, layer_idx=l).to(DEV).to(torch.bfloat16).eval() for l in range(nl)]
    g = torch.Generator().manual_seed(6)
    x = torch.randn(B, T, d, generator=g).to(torch.bfloat16).to(DEV)
    s = torch.randn(B, T, d, generator=g).to(torch.bfloat16).to(DEV)
    hidden = [torch.randn(B, Ltok, d, generator=g).to(torch.bfloat16).to(DEV) for _ in range(nl)]
    ln = torch.tensor(lengths, dtype=torch.int32, device=DEV)
    # offline: the whole utterances at once
    with torch.no_grad():
        scores = scorer(x).squeeze(-1).float()
        idx, counts, sg, mem_off = _offline_memory(ops, scores, s, lengths)
        kmax = int(counts.max())
        live = torch.arange(kmax, device=DEV)[None, :] < counts[:, None]
        amask = ((~live).float() * torch.finfo(torch.float32).min)[:, None, None, :]
        off_out = [m(hidden[l], encoder_hidden_states=mem_off[:, :kmax].contiguous(), encoder_attention_mask=amask,
                     scorer_logits=sg[:, :kmax].contiguous(), limit_cross_attn=True)[0] for l, m in enumerate(mods)]
    # streaming
    seg = StreamingNuggetSegmenter(scorer, B, d, max_nuggets=kmax + 5, device=DEV)
    cache = StreamingEncoderKV(mods, seg)
    written = [0] * B                                   # tokens decoded so far, per utterance
    outs = [[[None] * Ltok for _ in range(B)] for _ in range(nl)]
    off = 0
    with torch.no_grad():
        for c in [160] * 6:
            known = torch.where(ln <= off + c, ln, torch.full_like(ln, 2 ** 31 - 1))
            seg.push(x[:, off:off + c].contiguous(), s[:, off:off + c].contiguous(), lengths=known)
            off += c
            ready = seg.tokens_ready().cpu().numpy()                # the policy decision is a host decision
            cache.extend(seg.state.counts.cpu().tolist())
            kvs = cache.layers()
            emask = cache.encoder_attention_mask()
            _, msc, _, _ = seg.memory()
            # decode, token by token, every token that is ready for ALL rows of the batch step (a real decoder would
            # run per utterance; rows that are not ready are simply not compared)
            for b in range(B):
                while written[b] < min(int(ready[b]), Ltok):
                    i = written[b]
                    for l, m in enumerate(mods):
                        o = m(hidden[l][:, i:i + 1].contiguous(), encoder_kv=kvs[l], encoder_attention_mask=emask,
                              scorer_logits=msc, limit_cross_attn=True, query_offset=i)[0]
                        outs[l][b][i] = o[b, 0].float()
                    written[b] += 1
    assert written == [Ltok] * B and bool(seg.finished().all()) and int(seg.overflow().sum()) == 0
    for l in range(nl):
        ref = off_out[l].float()
        for b in range(B):
            got = torch.stack(outs[l][b])
            assert (got - ref[b]).abs().max() <= 2e-2 * ref[b].abs().max() + 1e-3, (l, b)


# ------------------------------------------------------------------------------------------------ f1: projections
@pytest.mark.parametrize("M,N,K,bias", [(12000, 8192, 1024, True), (1000, 2048, 1024, True), (300, 320, 64, False),
                                        (257, 64, 192, True), (1, 1024, 1024, True)])
def test_projection_gemm_vs_fp32_reference(ops, M, N, K, bias):
    """star_gemm_nt_bf16 (row a9 / 8f-1: y = x W + b of HF Conv1D, my_transformer.py:500-501,526) against the fp32
    product of the same bf16 operands: one rounding to bf16 plus fp32 accumulation noise.  The first shape is the
    c_attn of all four decoder layers on the configs[1] memory (the grouped XCD tile order), the others the edge cases:
    ragged row tile, N not a multiple of the tile, a single K tile, one row."""
    g = torch.Generator().manual_seed(M + N)
    a = torch.randn(M, K, generator=g).to(torch.bfloat16).to(DEV)
    wt = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16).to(DEV)
    b = torch.randn(N, generator=g).to(DEV) if bias else None
    out = ops.gemm_nt_bf16(a, wt, b)
    ref = a.float() @ wt.float().t()
    if bias:
        ref = ref + b
    assert out.shape == (M, N) and out.dtype == torch.bfloat16
    err = (out.float() - ref).abs()
    assert (err <= 2.0 ** -8 * ref.abs() + 1e-3).all(), err.max().item()
    # strided input (a column slice of a wider matrix: the K / V halves of a c_attn output are such views)
    wide = torch.randn(M, K + 64, generator=g).to(torch.bfloat16).to(DEV)
    out2 = ops.gemm_nt_bf16(wide[:, 64:], wt, b)
    ref2 = wide[:, 64:].float() @ wt.float().t() + (b if bias else 0)
    assert ((out2.float() - ref2).abs() <= 2.0 ** -8 * ref2.abs() + 1e-3).all()


@pytest.mark.parametrize("M,N,dtype", [(4096, 1024, torch.bfloat16), (12000, 8192, torch.bfloat16), (130, 72, torch.float32),
                                       (1, 8, torch.bfloat16)])
def test_colsum_bias_gradient(ops, M, N, dtype):
    """star_colsum (the projections' bias gradient): against an fp64 column sum, and bitwise reproducible."""
    g = torch.Generator().manual_seed(N)
    x = torch.randn(M, N, generator=g).to(dtype).to(DEV)
    s1, s2 = ops.colsum(x), ops.colsum(x)
    assert torch.equal(s1, s2) and s1.dtype == torch.float32
    ref = x.double().sum(0)
    assert (s1.double() - ref).abs().max() <= 1e-5 * x.double().abs().sum(0).max() + 1e-6
    xs = torch.randn(M, N + 8, generator=g).to(dtype).to(DEV)[:, 8:]              # a strided view
    assert (ops.colsum(xs).double() - xs.double().sum(0)).abs().max() <= 1e-5 * xs.double().abs().sum(0).max() + 1e-6


def test_conv1d_and_shared_projection_gradients(ops):
    """ops.conv1d (torch GEMMs + star_colsum bias gradient) and ops.shared_conv1d (all layers' c_attn in one launch of
    star_gemm_nt_bf16, no weight concatenation) against plain torch addmm autograd on the same bf16 operands."""
    from star_amd.nugget import NuggetCrossAttention, share_encoder_kv
    torch.manual_seed(9)
    B, Kn, d, h, nl = 32, 375, 1024, 8, 4
    mods = [NuggetCrossAttention(d, h, layer_idx=i).to(DEV, torch.bfloat16) for i in range(nl)]
    mem = torch.randn(B, Kn, d, device=DEV).to(torch.bfloat16).requires_grad_()
    gk = [torch.randn(B, h, Kn, d // h, device=DEV).to(torch.bfloat16) * 0.1 for _ in range(2 * nl)]
    kvs = share_encoder_kv(mods, mem)
    outs = [t for kv in kvs for t in (kv.key, kv.value)]
    torch.autograd.backward(outs, gk)
    got = {"mem": mem.grad.clone(), "w": [m.c_attn.weight.grad.clone() for m in mods],
           "b": [m.c_attn.bias.grad.clone() for m in mods], "k0": kvs[0].key.detach().clone(),
           "v3": kvs[3].value.detach().clone()}
    mem.grad = None
    for m in mods:
        m.c_attn.weight.grad = m.c_attn.bias.grad = None
    # reference: per-layer torch addmm on the same parameters
    ref_outs = []
    for m in mods:
        kv = torch.addmm(m.c_attn.bias, mem.view(-1, d), m.c_attn.weight).view(B, Kn, 2 * d)
        k, v = kv.split(d, dim=2)
        ref_outs += [k.view(B, Kn, h, d // h).permute(0, 2, 1, 3), v.view(B, Kn, h, d // h).permute(0, 2, 1, 3)]
    torch.autograd.backward(ref_outs, gk)

    def close(a, b, what, tol=2e-2):
        a, b = a.float(), b.float()
        assert (a - b).abs().max() <= tol * b.abs().max() + 1e-6, (what, (a - b).abs().max().item(), b.abs().max().item())
    close(got["k0"], ref_outs[0], "K of layer 0", 1e-2)
    close(got["v3"], ref_outs[7], "V of layer 3", 1e-2)
    close(got["mem"], mem.grad, "d mem")
    for l, m in enumerate(mods):
        close(got["w"][l], m.c_attn.weight.grad, f"dW layer {l}")
        close(got["b"][l], m.c_attn.bias.grad, f"db layer {l}")
    # a weight update is seen (the transposed copies are keyed by the parameters' version counters)
    with torch.no_grad():
        mods[1].c_attn.weight.mul_(2.0)
    kv2 = share_encoder_kv(mods, mem.detach())
    close(kv2[1].key, (torch.addmm(mods[1].c_attn.bias, mem.detach().view(-1, d), mods[1].c_attn.weight)
                       .view(B, Kn, 2 * d)[:, :, :d].view(B, Kn, h, d // h).permute(0, 2, 1, 3)), "K after an update", 1e-2)
    # the small projections: q_attn-shaped Conv1D, gradients against torch
    x = torch.randn(B, 128, d, device=DEV).to(torch.bfloat16).requires_grad_()
    gy = torch.randn(B, 128, d, device=DEV).to(torch.bfloat16)
    q = mods[0].q_attn
    q(x).backward(gy)
    gw, gb, gx = q.weight.grad.clone(), q.bias.grad.clone(), x.grad.clone()
    q.weight.grad = q.bias.grad = x.grad = None
    torch.addmm(q.bias, x.view(-1, d), q.weight).view(B, 128, d).backward(gy)
    close(gx, x.grad, "q_attn dx", 1e-2)
    close(gw, q.weight.grad, "q_attn dW", 1e-2)
    close(gb, q.bias.grad, "q_attn db", 1e-2)


@pytest.mark.parametrize("form", ["dense", "compact", "ragged"])
def test_scorer_forward_4wave_kernel_equals_8wave(ops, form):
    """The experimental 4-wave register-staged scorer GEMM (gemm_nt4.h, test override STAR_OVR_SCORER_FWD = 1) against
    the shipped 8-wave kernel: same accumulation order over k, so the ReLU sign bits -- and with them every gradient --
    are bit-identical; the scores differ only in the order of the 256-column row sums (tolerance 1e-5 of the scale).
    Backward through the 4-wave forward's bits is compared as well."""
    torch.manual_seed(11)
    B, T, d = 3, 700, 512
    lengths = [700, 333, 0] if form != "dense" else [T] * B
    valid = None if form == "dense" else ops.ValidFrames(lengths, T, DEV, compact=(form == "compact"))
    x = torch.randn(B, T, d, device=DEV).bfloat16()
    w1 = (torch.randn(d, d, device=DEV) / d ** 0.5)
    b1, w2, b2 = torch.randn(d, device=DEV) * 0.1, torch.randn(d, device=DEV) / d ** 0.5, torch.randn(1, device=DEV)
    gs = torch.randn(B, T, device=DEV)
    res = []
    try:
        for ov in (0, 1):
            ops.test_override(ops.OVR_SCORER_FWD, ov)
            xs = x.clone().requires_grad_()
            ps = [t.clone().requires_grad_() for t in (w1, b1, w2, b2)]
            s = ops.scorer(xs, *ps, None, valid)
            (s * gs).sum().backward()
            res.append((s.detach(), xs.grad, [p.grad for p in ps]))
    finally:
        ops.test_override(ops.OVR_SCORER_FWD, 0)
    (s8, dx8, g8), (s4, dx4, g4) = res
    m = torch.ones(B, T, dtype=torch.bool, device=DEV) if valid is None else \
        (torch.arange(T, device=DEV)[None, :] < torch.tensor(lengths, device=DEV)[:, None])
    scale = s8[m].abs().max().item()
    assert (s4[m] - s8[m]).abs().max().item() <= 1e-5 * scale
    assert torch.equal(dx4, dx8)
    for a, b in zip(g4, g8):
        assert torch.equal(a, b)


def test_dropout_state_survives_capture_and_reseed(ops):
    """The per-device {seed, offset} tensor of attention dropout is ONE tensor for the life of the process: a graph
    captured after init_dropout_state() draws a fresh mask at every replay, a re-seed (reset_dropout_state) is written
    into the same tensor and is therefore seen by the replays of a graph captured BEFORE it, and a state that does not
    exist yet cannot be created under capture (clear error instead of an illegal copy)."""
    from star_amd import _lib
    torch.manual_seed(5)
    B, h, L, K, hd, p = 2, 4, 64, 96, 64, 0.3
    q, k, v = (torch.randn(B, h, n, hd, device=DEV).bfloat16() for n in (L, K, K))

    def fwd():
        return ops.xattn(q, k, v, dropout_p=p)[0]

    ops.reset_dropout_state(1001)
    ops.init_dropout_state(DEV)
    eager = [fwd().clone() for _ in range(3)]                      # masks 0, 1, 2 of seed 1001
    assert not torch.equal(eager[0], eager[1])
    ops.reset_dropout_state(1001)
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        fwd()                                                       # warm-up outside the capture (offset 0 consumed)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=stream):
            out = fwd()
        # (capture does not execute: the offset is still 1)
        for want in (eager[1], eager[2]):
            graph.replay()
            torch.cuda.synchronize()
            assert torch.equal(out, want)
        ops.reset_dropout_state(1001)                               # re-seed AFTER the capture, in place
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, eager[0])
        ops.reset_dropout_state(2002)
        graph.replay()
        torch.cuda.synchronize()
        assert not any(torch.equal(out, e) for e in eager)
    # a device without a state: creating one under capture is refused
    saved = dict(ops._DropoutState._states)
    ops._DropoutState._states.clear()
    try:
        with torch.cuda.stream(stream):
            g2 = torch.cuda.CUDAGraph()
            msg = None
            with torch.cuda.graph(g2, stream=stream):
                try:
                    fwd()
                except _lib.StarNuggetError as e:
                    msg = str(e)
                keep = q + 1                                        # (the capture holds one node)
            assert msg is not None and "init_dropout_state" in msg, msg
            del keep
    finally:
        ops._DropoutState._states.clear()
        ops._DropoutState._states.update(saved)
        ops.reset_dropout_state(None)


@pytest.mark.parametrize("world,rank", [(1, 0), (8, 3)])
def test_full_size_simul_config4_shards(ops, world, rank):
    """BASELINE.json configs[4]: ONE global batch of 512 MuST-C-shaped utterances sharded by utterance (bench.py
    --config 4 draws it with mustc_lengths(512, 1500, 4321) and star_amd.dist.shard).  The whole simultaneous step on
    the 512 utterances one GPU gets at N = 1, and on the 64 that rank 3 of 8 gets, through the size-independent
    properties of the configs[2] test (resize row sums, fires ascending with the forced last fire and equal to the
    oracle on sampled rows, exact gathers, dS only at fired frames, masked probabilities exactly 0)."""
    import importlib.util, os
    from star_amd import dist as sd
    from tests.test_gpu_round2 import _simul_full_size
    spec = importlib.util.spec_from_file_location("star_bench", os.path.join(os.path.dirname(__file__), "..", "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    n, tl = bench.mustc_lengths(512, 1500, 4321)
    n, tl = sd.shard(n, rank, world), sd.shard(tl, rank, world)
    assert len(n) == 512 // world
    _simul_full_size(ops, len(n), 1500, n, tl, False, 40 + rank)


# ------------------------------------------------------------------------------------------------ head mask
@pytest.mark.parametrize("name", ["headmask_a", "headmask_ilk"])
def test_score_biased_attn_head_mask_golden(ops, name):
    """my_transformer.py:598-599: `attn_weights * head_mask` after the dropout, with the [1, h, 1, 1] per-head factors
    HF's get_head_mask hands to a block (reference run: tests/golden/gen_golden.py gen_head_mask).  The drop-in applies
    the factor to the core's output and returned weights; output, weights and every gradient (the head mask's too)
    against the fp32 oracle on the bf16-rounded operands (2e-2) and against the reference's own numbers (5e-2)."""
    import os
    from star_amd.nugget import score_biased_attn
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", name + ".npz"))
    q, k, v, sc, hm, gout = (torch.from_numpy(g[n]) for n in ("q", "k", "v", "scores", "head_mask", "gout"))
    hd = q.shape[-1]
    pad = lambda t: torch.nn.functional.pad(t, (0, 64 - hd))          # zero-padding the head dim changes nothing
    kmask = torch.from_numpy(lmask(g["key_lengths"], k.shape[2]))
    # the kernel scales by 1/sqrt(head dim) of what it is given (64 after the padding): the query carries the
    # difference, c = sqrt(64 / hd), and the oracle sees exactly the query the kernel sees, divided by c again
    c = (64 / hd) ** 0.5
    qr = (bf16_round(q * c) / c).requires_grad_()
    kr, vr = (bf16_round(t).requires_grad_() for t in (k, v))
    scr, hmr = sc.clone().requires_grad_(), hm.clone().requires_grad_()
    o_ref, p_ref = orc.xattn_core(qr, kr, vr, kmask, scr, bool(g["ilk"]), head_mask=hmr)
    (o_ref * bf16_round(gout)).sum().backward()

    qd, kd, vd = (pad(t).to(torch.bfloat16).to(DEV).requires_grad_() for t in (q * c, k, v))
    scd, hmd = sc.to(DEV).requires_grad_(), hm.to(DEV).requires_grad_()
    add = ((1.0 - kmask.float()) * torch.finfo(torch.float32).min).view(kmask.shape[0], 1, 1, -1).to(DEV)
    o, p = score_biased_attn(qd, kd, vd, add, hmd, scd, limit_cross_attn=bool(g["ilk"]),
                             output_attentions=True)
    o.backward(pad(gout).to(torch.bfloat16).to(DEV))
    f = lambda t: t.float().cpu()[..., :hd].numpy()
    tol = dict(rtol=2e-2, atol=2e-2)
    np.testing.assert_allclose(f(o.detach()), o_ref.detach().numpy(), **tol)
    np.testing.assert_allclose(p.detach().float().cpu().numpy(), p_ref.detach().numpy(), rtol=1e-3, atol=1e-5)
    np.testing.assert_allclose(f(qd.grad) * c, qr.grad.numpy(), **tol)
    np.testing.assert_allclose(f(kd.grad), kr.grad.numpy(), **tol)
    np.testing.assert_allclose(f(vd.grad), vr.grad.numpy(), **tol)
    np.testing.assert_allclose(scd.grad.cpu().numpy(), scr.grad.numpy(), **tol)
    np.testing.assert_allclose(hmd.grad.cpu().numpy(), hmr.grad.numpy(), rtol=3e-2, atol=3e-2)
    np.testing.assert_allclose(f(o.detach()), g["o"], rtol=5e-2, atol=5e-2)
    # a masked head returns exact zeros, in the output and in the weights
    dead = (hm.view(-1) == 0).nonzero().view(-1)
    assert dead.numel() and (o.detach()[:, dead.to(DEV)] == 0).all() and (p.detach()[:, dead.to(DEV)] == 0).all()
    with pytest.raises(NotImplementedError):
        score_biased_attn(qd, kd, vd, add, torch.ones(1, q.shape[1], q.shape[2], 1, device=DEV), scd)


# ------------------------------------------------------------------------------------------------ batched dW (8f-1)
@pytest.mark.parametrize("nb,ks,R,M,N,pad", [(3, 1, 256, 256, 384, 0), (8, 2, 4096, 1024, 1024, 0), (1, 4, 1024, 136, 128, 8),
                                             (2, 8, 512, 520, 264, 16), (16, 1, 64, 128, 128, 0)])
def test_gemm_tn_batched_vs_fp32(ops, nb, ks, R, M, N, pad):
    """star_gemm_tn_batched_bf16 (the dW = x^T g of HF Conv1D's backward for several layers at once,
    my_transformer.py:500-501,526) against fp64 matmuls of the same bf16 operands: fp32 accumulation, so 1e-4 of the
    result's scale; row strides wider than the rows, ragged last tiles, every r-range count."""
    g = torch.Generator(device=DEV).manual_seed(nb * 1000 + R)
    xs = [torch.randn(R, M + pad, device=DEV, generator=g).bfloat16()[:, :M] for _ in range(nb)]
    gs = [torch.randn(R, N + pad, device=DEV, generator=g).bfloat16()[:, :N] for _ in range(nb)]
    out = ops.gemm_tn_batched(xs, gs, ks)
    assert out.shape == (nb, M, N) and out.dtype == torch.float32
    for i in range(nb):
        ref = xs[i].double().t() @ gs[i].double()
        err = (out[i].double() - ref).abs().max().item()
        assert err <= 1e-4 * ref.abs().max().item(), (i, err)
    # the bias gradients that go with them: one launch pair, each row bit-identical to star_colsum of that matrix
    cs = ops.colsum_batched(gs)
    for i in range(nb):
        assert torch.equal(cs[i], ops.colsum(gs[i]))
    with pytest.raises(ValueError):
        ops.gemm_tn_batched(xs, gs[:-1] + [gs[-1][:, :N - 8]], ks)         # mixed shapes
    with pytest.raises(ValueError):
        ops.gemm_tn_batched([x[:R - 8] for x in xs], [t[:R - 8] for t in gs], ks)      # R not a multiple of 64


def test_deferred_projection_weight_grads_equal_immediate(ops):
    """defer_projection_weight_grads: q_attn / c_proj weight gradients formed in one launch when backward() ends
    (ops._DeferredWeightGrads) against the per-layer torch.mm gradients of the same stack: the weights' .grad equal up
    to the bf16 rounding of two fp32 sums taken in different orders, every other gradient bit-identical; a second
    backward accumulates; modules that do not fit the kernel (fp32 here) keep the immediate path."""
    from star_amd.nugget import NuggetCrossAttention, defer_projection_weight_grads
    torch.manual_seed(3)
    B, L, K, d, h, nl = 4, 64, 50, 256, 4, 3
    mods = [NuggetCrossAttention(d, h, layer_idx=i).to(DEV, torch.bfloat16) for i in range(nl)]
    mem = torch.randn(B, K, d, device=DEV).bfloat16().requires_grad_()
    xs = [torch.randn(B, L, d, device=DEV).bfloat16().requires_grad_() for _ in range(nl)]
    gouts = [torch.randn(B, L, d, device=DEV).bfloat16() for _ in range(nl)]
    sc = torch.randn(B, K, device=DEV).requires_grad_()

    def run(times=1):
        for t in [mem, sc] + xs + [p for m in mods for p in m.parameters()]:
            t.grad = None
        for _ in range(times):
            outs = [m(x, encoder_hidden_states=mem, scorer_logits=sc)[0] for m, x in zip(mods, xs)]
            torch.autograd.backward(outs, gouts)
        torch.cuda.synchronize()
        return ({n + str(i): p.grad.clone() for i, m in enumerate(mods) for n, p in m.named_parameters()},
                [t.grad.clone() for t in [mem, sc] + xs])

    w_ref, o_ref = run()
    assert defer_projection_weight_grads(mods) == nl
    w_def, o_def = run()
    w_def2, _ = run(times=2)
    defer_projection_weight_grads(mods, False)
    w_off, _ = run()
    for a, b in zip(o_ref, o_def):
        assert torch.equal(a, b)
    for n in w_ref:
        if "q_attn.weight" in n or "c_proj.weight" in n:
            scale = w_ref[n].float().abs().max().item()
            assert (w_def[n].float() - w_ref[n].float()).abs().max().item() <= 1e-2 * scale, n
            assert (w_def2[n].float() - 2 * w_ref[n].float()).abs().max().item() <= 3e-2 * scale, n
            assert not torch.equal(w_def[n], torch.zeros_like(w_def[n]))
        else:
            assert torch.equal(w_def[n], w_ref[n]), n
        assert torch.equal(w_off[n], w_ref[n]), n
    # fp32 modules do not fit the bf16 kernel: the flag is accepted and the gradients come the usual way
    m32 = NuggetCrossAttention(d, h).to(DEV)
    defer_projection_weight_grads(m32)
    x32 = torch.randn(2, 8, d, device=DEV, requires_grad=True)
    m32(x32.bfloat16().float(), encoder_hidden_states=mem.detach().float())[0].sum().backward()
    assert m32.q_attn.weight.grad is not None and m32.c_proj.weight.grad is not None


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_transpose_cat_bf16_equals_torch(ops, dtype):
    """star_transpose_list_bf16 (the all-layer c_attn weight of share_encoder_kv, rebuilt once per optimizer step)
    against cat(dim=1).t().contiguous() of the bf16-cast weights: bit-identical."""
    g = torch.Generator(device=DEV).manual_seed(17)
    ws = [torch.randn(192, 320, device=DEV, generator=g).to(dtype) for _ in range(5)]
    got = ops.transpose_cat_bf16(ws)
    want = torch.cat([w.to(torch.bfloat16) for w in ws], dim=1).t().contiguous()
    assert got.shape == want.shape and torch.equal(got, want)
    with pytest.raises(ValueError):
        ops.transpose_cat_bf16([w[:, :200] for w in ws])


@pytest.mark.parametrize("name", ["scaling_noscale", "scaling_invlayer"])
def test_score_biased_attn_scaling_switches_golden(ops, name):
    """scale_attn_weights=False and scale_attn_by_inverse_layer_idx=True with layer_idx = 2 (my_transformer.py:538-545)
    through score_biased_attn's own arguments, against the oracle on the bf16-rounded operands (2e-2) and the
    reference's numbers (5e-2); reference run: tests/golden/gen_golden.py gen_xattn_scaling."""
    import os
    from star_amd.nugget import score_biased_attn
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", name + ".npz"))
    q, k, v, sc, gout = (torch.from_numpy(g[n]) for n in ("q", "k", "v", "scores", "gout"))
    saw, inv, li, ilk = bool(g["scale_attn_weights"]), bool(g["inverse_layer_idx"]), int(g["layer_idx"]), bool(g["ilk"])
    hd = q.shape[-1]
    pad = lambda t: torch.nn.functional.pad(t, (0, 64 - hd))
    c = (64 / hd) ** 0.5 if saw else 1.0          # the kernel sees head dim 64 after the zero padding
    kmask = torch.from_numpy(lmask(g["key_lengths"], k.shape[2]))
    qr = (bf16_round(q * c) / c).requires_grad_()
    kr, vr = (bf16_round(t).requires_grad_() for t in (k, v))
    scr = sc.clone().requires_grad_()
    o_ref, p_ref = orc.xattn_core(qr, kr, vr, kmask, scr, ilk, scale=saw, inverse_layer_idx=li if inv else None)
    (o_ref * bf16_round(gout)).sum().backward()
    qd, kd, vd = (pad(t).to(torch.bfloat16).to(DEV).requires_grad_() for t in (q * c, k, v))
    scd = sc.to(DEV).requires_grad_()
    add = ((1.0 - kmask.float()) * torch.finfo(torch.float32).min).view(kmask.shape[0], 1, 1, -1).to(DEV)
    o, p = score_biased_attn(qd, kd, vd, add, None, scd, limit_cross_attn=ilk, scale_attn_weights=saw,
                             scale_attn_by_inverse_layer_idx=inv, layer_idx=li, output_attentions=True)
    o.backward(pad(gout).to(torch.bfloat16).to(DEV))
    f = lambda t: t.float().cpu()[..., :hd].numpy()
    tol = dict(rtol=2e-2, atol=2e-2)
    np.testing.assert_allclose(f(o.detach()), o_ref.detach().numpy(), **tol)
    np.testing.assert_allclose(p.detach().float().cpu().numpy(), p_ref.detach().numpy(), rtol=1e-3, atol=1e-5)
    np.testing.assert_allclose(f(qd.grad) * c, qr.grad.numpy(), **tol)
    np.testing.assert_allclose(f(kd.grad), kr.grad.numpy(), **tol)
    np.testing.assert_allclose(f(vd.grad), vr.grad.numpy(), **tol)
    np.testing.assert_allclose(scd.grad.cpu().numpy(), scr.grad.numpy(), **tol)
    np.testing.assert_allclose(f(o.detach()), g["o"], rtol=5e-2, atol=5e-2)


def test_deferred_projection_weight_grads_under_graph_capture(ops):
    """The end-of-backward callback of defer_projection_weight_grads runs inside a HIP-graph capture like the rest of
    the backward pass (tools/f1_stack.py times it that way): a captured forward + backward, replayed on new inputs,
    leaves the gradients of an eager run on those inputs in the captured .grad tensors."""
    from star_amd.nugget import NuggetCrossAttention, defer_projection_weight_grads
    torch.manual_seed(11)
    B, L, K, d, h, nl = 2, 64, 40, 256, 4, 2
    mods = [NuggetCrossAttention(d, h, layer_idx=i).to(DEV, torch.bfloat16) for i in range(nl)]
    defer_projection_weight_grads(mods)
    mem = torch.randn(B, K, d, device=DEV).bfloat16()
    x = torch.randn(B, L, d, device=DEV).bfloat16()
    gout = torch.randn(B, L, d, device=DEV).bfloat16()
    params = [p for m in mods for p in m.parameters()]

    def step():
        for p in params:
            p.grad = None
        outs = [m(x, encoder_hidden_states=mem)[0] for m in mods]
        torch.autograd.backward(outs, [gout] * nl)

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step()
    captured = [p.grad for p in params]
    assert all(g is not None for g in captured)
    for trial in range(2):
        x.copy_(torch.randn(B, L, d, device=DEV).bfloat16())
        mem.copy_(torch.randn(B, K, d, device=DEV).bfloat16())
        graph.replay()
        torch.cuda.synchronize()
        got = [g.clone() for g in captured]
        step()                                    # eager, same inputs (rebinds .grad: `captured` keeps the graph's)
        torch.cuda.synchronize()
        for p, a in zip(params, got):
            assert torch.equal(a, p.grad), trial
