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
    close(scorer[2].bias.grad, b2.grad, "db2")
    close(inp.s.grad, sr.grad, "dS")
    close(inp.x.grad, xr.grad, "dX")
    for l in range(nl):
        close(inp.qs[l].grad, qs[l].grad, f"dQ{l}")
