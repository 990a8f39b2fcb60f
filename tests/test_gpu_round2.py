"""GPU parity, round-2 additions (all through the C ABI): attention-probability dropout, the prefix-bias variant
(bias_len < Kn), resize backward, get_valid_mask_indices, gather edge cases, the single-use active-row hint.

Bars as in tests/test_gpu_parity.py: integer results bit-exact; bf16-MFMA results against the fp32 oracle on the same
bf16-rounded inputs at 2e-2 of the tensor scale; fp32 scalar paths rtol 1e-5."""
import numpy as np
import pytest
import torch

from oracle import nugget_oracle as orc
from tests.test_oracle_golden import lmask, load

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    from star_amd import ops as o
    return o


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)) if isinstance(a, np.ndarray) else a
    if dtype is not None:
        t = t.to(dtype)
    return t.to(DEV)


def bf16_round(t):
    return t.to(torch.bfloat16).to(torch.float32)


def close(a, b, what, tol=2e-2):
    a = a.detach().float().cpu().numpy()
    b = b.detach().float().cpu().numpy()
    scale = np.abs(b).max()
    assert np.abs(a - b).max() <= tol * scale + 1e-6, f"{what}: max err {np.abs(a - b).max()} scale {scale}"


# ------------------------------------------------------------------------------------------------ ADVICE r1 (high)
def test_get_valid_mask_indices_golden():
    """Every returned value of the mirrored method against the reference golden (w2v_simul.py:196-220): masks whose
    last element is False must not get a forced last-frame index; 1-D and 2-D; non-default padding value."""
    from star_amd.nugget import NuggetOpsMixin
    g = load("valid_indices")
    for tag in "abcd":
        m = dev(g[f"mask_{tag}"])
        idx, kmax, n = NuggetOpsMixin.get_valid_mask_indices(m, padding_value=int(g[f"pad_{tag}"]))
        np.testing.assert_array_equal(idx.cpu().numpy(), g[f"idx_{tag}"])
        assert kmax == int(g[f"kmax_{tag}"])
        np.testing.assert_array_equal(n.cpu().numpy(), g[f"n_{tag}"])
        assert idx.dtype == torch.int64


def test_get_valid_mask_indices_random(ops):
    from star_amd.nugget import NuggetOpsMixin
    rng = np.random.default_rng(5)
    for B, T in ((7, 1500), (3, 257), (1, 16), (4, 48)):
        m = rng.random((B, T)) < 0.07
        m[0, -1] = False
        m[-1, -1] = True
        m[B // 2, :] = False          # an empty row
        o_idx, o_k, o_n = orc.valid_mask_indices(m, -1)
        idx, kmax, n = NuggetOpsMixin.get_valid_mask_indices(dev(m))
        assert kmax == o_k
        np.testing.assert_array_equal(idx.cpu().numpy(), o_idx)
        np.testing.assert_array_equal(n.cpu().numpy(), o_n)


def test_fire_scan_forced_fire_only_inside_the_row(ops):
    """lengths > T: every frame valid, no forced fire (the star_cif_fire_scan contract get_valid_mask_indices uses);
    lengths == T: forced fire at T-1 as before."""
    a = torch.full((2, 10), 0.26, device=DEV)          # integral: 1.04 at t = 3 (fire), 1.08 at t = 7 (fire), 0.60 at t = 9
    idx, _, counts, _, _ = ops.cif_fire_scan(a, lengths=torch.tensor([10, 11], dtype=torch.int32, device=DEV))
    c = counts.cpu().numpy()
    i = idx.cpu().numpy()
    assert list(i[0, :c[0]]) == [3, 7, 9] and list(i[1, :c[1]]) == [3, 7]     # the forced fire at 9 only in row 0


# ------------------------------------------------------------------------------------------------ resize backward
@pytest.mark.parametrize("name", ["resize_grad_plain", "resize_grad_fixup"])
def test_resize_backward_golden(ops, name):
    g = load(name)
    a = dev(g["alphas"]).requires_grad_()
    r = ops.cif_resize(a, dev(g["target_lengths"], torch.int32))
    # forward as in round 1: fp64 row sums rounded once vs torch's fp32 sums
    np.testing.assert_allclose(r.detach().cpu().numpy(), g["resized"], rtol=2e-5, atol=1e-7)
    (r * dev(g["gout"])).sum().backward()
    np.testing.assert_allclose(a.grad.cpu().numpy(), g["dalphas"], rtol=1e-4, atol=1e-6)


def test_resize_backward_random_and_mixin(ops):
    """Random rows (some needing fix-ups) against the oracle's autograd; the mirrored NuggetOpsMixin.resize carries
    the gradient (lm_cif_training trains the scorer through it, w2v_simul.py:558)."""
    from star_amd.nugget import NuggetOpsMixin
    gen = torch.Generator().manual_seed(3)
    B, T = 9, 300
    scores = torch.randn(B, T, generator=gen) - 1.0
    scores[2, 17] = 9.0
    scores[5, :] = -5.0
    scores[5, 100] = 6.0
    lengths = [300, 280, 40, 129, 64, 200, 7, 1, 300]
    tgt = torch.tensor([30, 25, 9, 12, 6, 14, 3, 1, 2])
    mask = torch.from_numpy(lmask(lengths, T))
    alphas = torch.sigmoid(scores) * mask
    gout = torch.randn(B, T, generator=gen)
    ar = alphas.clone().requires_grad_()
    r_ref = orc.resize(ar, tgt, differentiable=True)
    (r_ref * gout).sum().backward()
    ad = alphas.to(DEV).requires_grad_()
    r = NuggetOpsMixin().resize(ad, tgt.to(DEV))
    np.testing.assert_allclose(r.detach().cpu().numpy(), r_ref.detach().numpy(), rtol=2e-5, atol=1e-7)
    (r * gout.to(DEV)).sum().backward()
    np.testing.assert_allclose(ad.grad.cpu().numpy(), ar.grad.numpy(), rtol=2e-4, atol=2e-6)
    assert (ad.grad.cpu().numpy() != 0).any()


# ------------------------------------------------------------------------------------------------ attention dropout
def _drop_case(B, h, L, K, hd, seed):
    g = torch.Generator().manual_seed(seed)
    q = torch.randn(B, h, L, hd, generator=g)
    k = torch.randn(B, h, K, hd, generator=g)
    v = torch.randn(B, h, K, hd, generator=g)
    sc = torch.randn(B, K, generator=g)
    gout = torch.randn(B, h, L, hd, generator=g)
    klen = [K] + [max(3, K - 37 * (i + 1)) for i in range(B - 1)]
    return q, k, v, sc, gout, klen


DROP_SHAPES = [(2, 8, 128, 375, 128, False, 0.1),      # configs[1]-shaped: fused backward
               (2, 4, 24, 63, 96, False, 0.1),         # hd 96: general dQ / dK,dV kernels
               (1, 8, 227, 229, 128, True, 0.1),       # configs[3]-shaped, ILK, L > 128: ring dQ + register dK/dV
               (3, 2, 130, 200, 64, True, 0.25),
               (26, 8, 128, 130, 64, True, 0.1),       # fused, hd 64, ILK
               (2, 2, 60, 200, 64, False, 0.5)]


@pytest.mark.parametrize("B,h,L,K,hd,ilk,p", DROP_SHAPES)
def test_xattn_dropout_vs_oracle_same_mask(ops, B, h, L, K, hd, ilk, p):
    """O, P (dropped + rescaled, as the reference returns it), dQ, dK, dV, dscores against the oracle fed the SAME keep
    mask (restated from the specification in star_nugget.h); the kernel's mask equals that restatement bit for bit
    (read off the zeros of p_out); every backward variant regenerates it."""
    q, k, v, sc, gout, klen = _drop_case(B, h, L, K, hd, 1000 + L + K)
    seed, offset = 0x1234_5678_9ABC_DEF1, 42 + L
    keep, p_eff = orc.dropout_keep_mask(seed, offset, B, h, L, K, p)
    n = keep.size
    assert abs(keep.mean() - (1 - p_eff)) < 3 * (p_eff * (1 - p_eff) / n) ** 0.5 + 1e-9        # keep rate within 3 sigma
    qr, kr, vr = (bf16_round(t).requires_grad_() for t in (q, k, v))
    scr = sc.clone().requires_grad_()
    kmask = torch.from_numpy(lmask(klen, K))
    o_ref, p_ref = orc.xattn_core(qr, kr, vr, kmask, scr, ilk, drop_mask=torch.from_numpy(keep), drop_p=p_eff)
    (o_ref * bf16_round(gout)).sum().backward()
    key_len = torch.tensor(klen, dtype=torch.int32, device=DEV)
    variants = [[]]
    if hd in (64, 128):
        variants += [[(ops.OVR_XA_BWD, 1)], [(ops.OVR_XA_DQ, 1), (ops.OVR_XA_DKV, 1)]]
        if L <= 128:
            variants += [[(ops.OVR_XA_BWD, 2)], [(ops.OVR_XA_BWD, 3)]]     # 8-wave fused; head dim 128: the 4-wave form
    try:
        for switches in variants:
            for w, val in switches:
                ops.test_override(w, val)
            qd, kd, vd = (t.to(torch.bfloat16).to(DEV).requires_grad_() for t in (q, k, v))
            scd = sc.to(DEV).requires_grad_()
            o, pw = ops.xattn(qd, kd, vd, bias=scd, key_len=key_len, ilk_diag=2 if ilk else -1, need_probs=True,
                              dropout_p=p, dropout_rng=(seed, offset))
            o.backward(gout.to(torch.bfloat16).to(DEV))
            pk = pw.cpu().numpy()
            seen = np.broadcast_to(kmask.numpy()[:, None, None, :], keep.shape).copy()
            if ilk:
                seen &= (np.arange(K)[None, :] <= np.arange(L)[:, None] + 2)[None, None]
            # the kernel's zero pattern is exactly the oracle's mask (exp never underflows at these scales)
            np.testing.assert_array_equal((pk != 0)[seen], keep[seen])
            assert (pk[~seen] == 0).all()
            close(pw, p_ref, "P", 2e-3)
            close(o, o_ref, "O")
            close(qd.grad, qr.grad, "dQ")
            close(kd.grad, kr.grad, "dK")
            close(vd.grad, vr.grad, "dV")
            close(scd.grad, scr.grad, "dscores")
            for w, _ in switches:
                ops.test_override(w, 0)
    finally:
        for w in (ops.OVR_XA_BWD, ops.OVR_XA_DQ, ops.OVR_XA_DKV, ops.OVR_XA_NG):
            ops.test_override(w, 0)


def test_xattn_dropout_p0_is_bit_identical_and_state_advances(ops):
    q, k, v, sc, gout, klen = _drop_case(3, 4, 64, 140, 128, 77)
    qd, kd, vd = (t.to(torch.bfloat16).to(DEV) for t in (q, k, v))
    scd = sc.to(DEV)
    o0, _ = ops.xattn(qd, kd, vd, bias=scd)
    o1, _ = ops.xattn(qd, kd, vd, bias=scd, dropout_p=0.0, dropout_rng=(1, 2))
    assert torch.equal(o0, o1)
    # per-device state: consecutive calls draw different masks, the backward regenerates the forward's mask
    ops.reset_dropout_state(123)
    qg = qd.clone().requires_grad_()
    oa, pa = ops.xattn(qg, kd, vd, bias=scd, need_probs=True, dropout_p=0.3)
    ob, pb = ops.xattn(qd, kd, vd, bias=scd, need_probs=True, dropout_p=0.3)
    ka, kb = (pa != 0).cpu().numpy(), (pb != 0).cpu().numpy()
    assert abs(ka.mean() - 0.7) < 0.01 and abs(kb.mean() - 0.7) < 0.01 and (ka != kb).mean() > 0.2
    # gradient of the first call against the oracle with the mask read back from its p_out
    qr = bf16_round(q).requires_grad_()
    o_ref, _ = orc.xattn_core(qr, bf16_round(k), bf16_round(v), None, sc, False, drop_mask=torch.from_numpy(ka),
                              drop_p=0.3)
    (o_ref * bf16_round(gout)).sum().backward()
    oa.backward(gout.to(torch.bfloat16).to(DEV))
    close(oa, o_ref, "O")
    close(qg.grad, qr.grad, "dQ")
    # re-seeding reproduces the stream
    ops.reset_dropout_state(123)
    _, pc = ops.xattn(qd, kd, vd, bias=scd, need_probs=True, dropout_p=0.3)
    assert torch.equal(pa, pc)


def test_cross_attention_module_trains_with_default_attn_pdrop(ops):
    """The reference's default GPT2Config has attn_pdrop = 0.1 (w2v_asr.py:134-145): the module accepts it, applies
    dropout in training mode only, and gradients reach every projection."""
    from star_amd.nugget import NuggetCrossAttention
    torch.manual_seed(5)
    d, h, B, L, K = 256, 4, 3, 20, 50
    m = NuggetCrossAttention(d, h, attn_pdrop=0.1).to(DEV)
    x = torch.randn(B, L, d, device=DEV)
    mem = torch.randn(B, K, d, device=DEV)
    sl = torch.randn(B, K, device=DEV, requires_grad=True)
    m.eval()
    with torch.no_grad():
        e1 = m(x, encoder_hidden_states=mem, scorer_logits=sl)[0]
        e2 = m(x, encoder_hidden_states=mem, scorer_logits=sl)[0]
    assert torch.equal(e1, e2)
    m.train()
    t1, _, w = m(x, encoder_hidden_states=mem, scorer_logits=sl, output_attentions=True)
    assert not torch.equal(t1, e1)
    kept = (w != 0).float().mean().item()
    assert abs(kept - 0.9) < 0.02
    t1.square().mean().backward()
    for name, prm in m.named_parameters():
        assert prm.grad is not None and torch.isfinite(prm.grad).all() and prm.grad.abs().sum() > 0, name
    assert sl.grad is not None and sl.grad.abs().sum() > 0


# ------------------------------------------------------------------------------------------------ a11: prefix bias
@pytest.mark.parametrize("B,h,L,K,Kb,hd,ilk", [(2, 8, 128, 375, 200, 128, False), (3, 4, 40, 130, 1, 64, False),
                                               (1, 8, 227, 229, 64, 128, True), (2, 2, 24, 63, 31, 96, False),
                                               (26, 8, 100, 150, 77, 64, True)])
def test_prefix_bias_shorter_than_keys(ops, B, h, L, K, Kb, hd, ilk):
    """MyLlamaAttention's bias block (my_transformer.py:114-125): nugget logits cover only the first Kb < K keys and
    are zero-extended.  The kernels take bias_len = Kb directly (no padded copy); d(bias) has Kb columns."""
    from star_amd.nugget import score_biased_attn
    q, k, v, _, gout, klen = _drop_case(B, h, L, K, hd, 9 + Kb)
    sc = torch.randn(B, Kb, generator=torch.Generator().manual_seed(Kb))
    qr, kr, vr = (bf16_round(t).requires_grad_() for t in (q, k, v))
    scr = sc.clone().requires_grad_()
    kmask = torch.from_numpy(lmask(klen, K))
    o_ref, p_ref = orc.xattn_core(qr, kr, vr, kmask, scr, ilk)
    (o_ref * bf16_round(gout)).sum().backward()
    key_len = torch.tensor(klen, dtype=torch.int32, device=DEV)
    for via_module_fn in (False, True):
        qd, kd, vd = (t.to(torch.bfloat16).to(DEV).requires_grad_() for t in (q, k, v))
        scd = sc.to(DEV).requires_grad_()
        if via_module_fn:      # the drop-in _attn signature; key lengths stand in for the additive pad mask
            o, pw = score_biased_attn(qd, kd, vd, None, None, scd, limit_cross_attn=ilk, key_lengths=key_len,
                                      output_attentions=True)
        else:
            o, pw = ops.xattn(qd, kd, vd, bias=scd, key_len=key_len, ilk_diag=2 if ilk else -1, need_probs=True)
        o.backward(gout.to(torch.bfloat16).to(DEV))
        assert scd.grad.shape == (B, Kb)
        close(pw, p_ref, "P", 2e-3)
        close(o, o_ref, "O")
        close(qd.grad, qr.grad, "dQ")
        close(kd.grad, kr.grad, "dK")
        close(vd.grad, vr.grad, "dV")
        close(scd.grad, scr.grad, "dscores")


# ------------------------------------------------------------------------------------------------ gather edge cases
def test_gather_bf16_scores_and_out_of_range_clamp(ops):
    """2-D bf16 scores (row_bytes = 2: the d = 1 score gather of w2v_simul.py:427 under bf16); an index outside
    [-T, T) is clamped identically in forward and backward."""
    T = 31
    s = torch.randn(3, T).to(torch.bfloat16)
    idx = torch.tensor([[0, 5, 30, -1], [3, -1, -1, -31], [1, 2, 2, 4]])
    sd = s.to(DEV).requires_grad_()
    out = ops.gather_rows(sd, idx.to(DEV))
    assert torch.equal(out.detach().cpu(), s[torch.arange(3).unsqueeze(1), idx])
    w = torch.randn(3, 4).to(torch.bfloat16)
    out.backward(w.to(DEV))
    acc = torch.zeros(3, T)
    for b in range(3):
        for j in range(4):
            acc[b, idx[b, j].item()] += w[b, j].float()
    assert torch.equal(sd.grad.cpu(), acc.to(torch.bfloat16))
    # out of range: T + 5 -> frame T-1, -T - 3 -> frame 0, in both directions, for rows and for scalars
    bad = torch.tensor([[T + 5, -T - 3, 7]])
    for src in (torch.randn(1, T, 64), torch.randn(1, T)):
        srcd = src.to(DEV).requires_grad_()
        o = ops.gather_rows(srcd, bad.to(DEV))
        ref = src[:, [T - 1, 0, 7]]
        assert torch.equal(o.detach().cpu(), ref)
        o.backward(torch.ones_like(o))
        gsum = srcd.grad.cpu().reshape(T, -1)[:, 0]
        assert gsum[T - 1] == 1 and gsum[0] == 1 and gsum[7] == 1 and gsum.sum() == 3
    with pytest.raises(TypeError):
        ops.gather_rows(torch.zeros(1, 4, 8, dtype=torch.float16, device=DEV), bad.to(DEV))


# ------------------------------------------------------------------------------------------------ scorer hint
def test_scorer_hint_does_not_leak_into_later_backward(ops):
    """after a step that used the sparse-row hint, a dense loss through the same scorer must get the
    full gradient (the hint is single-use), and a hint set after a SECOND forward only binds to that forward."""
    from star_amd.nugget import NuggetScorer
    torch.manual_seed(0)
    d, N = 256, 260
    sc = NuggetScorer(d).to(DEV)
    x = torch.randn(2, N // 2, d, device=DEV).to(torch.bfloat16)
    gs = torch.randn(2, N // 2, device=DEV)

    def grads(set_rows):
        sc.zero_grad()
        s = sc(x).squeeze(-1)
        if set_rows is not None:
            sc.set_active_rows(set_rows)
        (s * gs).sum().backward()
        return [p.grad.clone() for p in sc.parameters()]

    dense = grads(None)
    rows = torch.arange(0, N, 5, dtype=torch.int32, device=DEV)
    sparse = grads(rows)
    assert not torch.equal(sparse[0], dense[0])            # only the listed rows contributed
    again = grads(None)                                    # the next step forgets the hint: dense, not stale
    for a, b in zip(again, dense):
        assert torch.equal(a, b)
    with pytest.raises(ValueError):
        grads(torch.arange(N + 1, dtype=torch.int32, device=DEV))


# ------------------------------------------------------------------------------------------------ forward variants
@pytest.mark.parametrize("B,h,L,K,hd,ilk", [(4, 8, 128, 375, 128, False), (2, 4, 300, 750, 128, True),
                                            (3, 2, 60, 63, 64, False), (2, 8, 227, 229, 128, True),
                                            (1, 2, 16, 1000, 64, False)])
def test_xattn_forward_ring_vs_register_staged(ops, B, h, L, K, hd, ilk):
    """The LDS-DMA ring forward (default for hd 64 / 128) and the register-staged forward (hd 96; forced here through
    star_test_override) against the oracle and against each other, with ragged key lengths and without a bias."""
    q, k, v, sc, gout, klen = _drop_case(B, h, L, K, hd, 31 + K)
    kmask = torch.from_numpy(lmask(klen, K))
    key_len = torch.tensor(klen, dtype=torch.int32, device=DEV)
    qd, kd, vd = (t.to(torch.bfloat16).to(DEV) for t in (q, k, v))
    for use_bias in (True, False):
        o_ref, p_ref = orc.xattn_core(bf16_round(q), bf16_round(k), bf16_round(v), kmask, sc if use_bias else None, ilk)
        outs = []
        try:
            for mode in (0, 1):
                ops.test_override(ops.OVR_XA_FWD, mode)
                o, pw = ops.xattn(qd, kd, vd, bias=sc.to(DEV) if use_bias else None, key_len=key_len,
                                  ilk_diag=2 if ilk else -1, need_probs=True)
                close(o, o_ref, f"O mode {mode}")
                close(pw, p_ref, f"P mode {mode}", 2e-3)       # P = exp(logit - lse): checks the LSE of either kernel
                outs.append(o.float().cpu())
        finally:
            ops.test_override(ops.OVR_XA_FWD, 0)
        close(outs[0], outs[1], "ring vs register-staged", 1e-2)


# ------------------------------------------------------------------------------------------------ whole simul step
@pytest.mark.parametrize("ilk", [False, True])
def test_nugget_step_simul_gradients_vs_oracle(ops, ilk):
    """The simultaneous NuggetStep against the fp32 oracle step (w2v_simul.py:407-461): scores, alphas, quantity loss,
    resized alphas, fire positions (bit-exact on the kernel's own resized alphas), loss, and every gradient of the
    step -- scorer parameters (quantity loss + attention bias through the score gather), S, Q."""
    from star_amd.nugget import NuggetScorer
    from star_amd.step import make_inputs, nugget_step_simul
    torch.manual_seed(1)
    lengths, T, d, h, nl = [300, 280, 150, 60], 300, 256, 2, 2
    tl = [int(round(0.075 * n)) + 2 for n in lengths]
    L = max(tl)
    scorer = NuggetScorer(d).to(DEV)
    with torch.no_grad():
        scorer[2].bias.fill_(-2.0)
    inp = make_inputs(lengths, T, d, h, L, nl, DEV, target_lengths=tl)
    out = nugget_step_simul(scorer, inp, use_ilk=ilk, with_loss=True)
    mask = lmask(lengths, T)
    o_mask, o_idx, _ = orc.fire_scan(out["resized"].cpu().numpy(), mask, with_block_mask=False)
    kf = o_idx.shape[1]
    np.testing.assert_array_equal(out["indices"][:, :kf].cpu().numpy(), o_idx)
    assert (out["indices"][:, kf:] == -1).all()
    np.testing.assert_array_equal(out["counts"].cpu().numpy(), o_mask.sum(1))

    cpu = lambda t: t.detach().float().cpu()
    w1, b1, w2, b2 = (cpu(p).requires_grad_() for p in (scorer[0].weight, scorer[0].bias, scorer[2].weight, scorer[2].bias))
    w1r = bf16_round(w1.detach()).requires_grad_()
    xr, sr = cpu(inp.x).requires_grad_(), cpu(inp.s).requires_grad_()
    qs = [cpu(q).requires_grad_() for q in inp.qs]
    K = out["indices"].shape[1]
    fires = (out["nugget_mask"].cpu().numpy().astype(bool), out["indices"].cpu().numpy())
    ref = orc.nugget_step_simul(xr, sr, lengths, tl, ilk, w1r, b1, w2, b2, qs, [(cpu(a), cpu(b)) for a, b in inp.kv_scales],
                                [cpu(g) for g in inp.gs], fires=fires)
    assert fires[1].shape[1] == K
    np.testing.assert_allclose(out["scores"].detach().cpu().numpy()[mask], ref["scores"].detach().numpy()[mask],
                               rtol=2e-3, atol=2e-3)
    assert (out["scores"].detach().cpu().numpy()[~mask] == 0).all()          # ragged scorer: pad frames are 0
    np.testing.assert_allclose(out["alphas"].detach().cpu().numpy(), ref["alphas"].detach().numpy(), rtol=2e-3, atol=1e-4)
    np.testing.assert_allclose(out["resized"].detach().cpu().numpy(), ref["resized"].detach().numpy(), rtol=2e-3, atol=1e-4)
    assert abs(out["quantity_loss"].item() - ref["quantity_loss"].item()) <= 5e-3 * abs(ref["quantity_loss"].item()) + 1e-4
    ref["loss"].backward()
    assert abs(out["loss"].item() - ref["loss"].item()) <= 2e-2 * abs(ref["loss"].item()) + 1e-2
    close(scorer[0].weight.grad, w1r.grad, "dW1", 3e-2)
    close(scorer[2].weight.grad, w2.grad, "dw2", 3e-2)
    close(scorer[0].bias.grad, b1.grad, "db1", 3e-2)
    close(scorer[2].bias.grad, b2.grad, "db2", 3e-2)
    close(inp.s.grad, sr.grad, "dS", 3e-2)
    for l in range(nl):
        close(inp.qs[l].grad, qs[l].grad, f"dQ{l}", 3e-2)
    assert inp.x.grad is None and xr.grad is None          # w2v_simul.py:410: the scorer sees detached states


def test_two_stage_backward_equals_single_backward(ops):
    """bench.py's multi-GPU step splits the backward after the attention kernels (star_amd.step.attention_backward /
    backward_rest).  Same kernels in the same order on the same operands: bit-identical gradients."""
    from star_amd.nugget import NuggetScorer
    from star_amd.step import (attention_backward, attention_backward_layer, backward_rest, backward_rest_layers,
                               make_inputs, nugget_step_asr, nugget_step_simul)
    torch.manual_seed(2)
    lengths, T, d, h, nl = [200, 180, 77], 200, 256, 4, 3
    tl = [int(round(0.075 * n)) + 2 for n in lengths]
    scorer = NuggetScorer(d).to(DEV)
    inp = make_inputs(lengths, T, d, h, max(tl), nl, DEV, target_lengths=tl)
    leaves = list(scorer.parameters()) + [inp.x, inp.s] + inp.qs

    def grads():
        g = [None if t.grad is None else t.grad.clone() for t in leaves]
        for t in leaves:
            t.grad = None
        return g

    for simul in (False, True):
        if simul:
            nugget_step_simul(scorer, inp, use_ilk=True)
        else:
            nugget_step_asr(scorer, inp, 4)
        one = grads()
        out = nugget_step_simul(scorer, inp, use_ilk=True, backward=False) if simul else \
            nugget_step_asr(scorer, inp, 4, backward=False)
        dmem, dsg = attention_backward(out, inp)
        if simul:
            ql = out["quantity_loss"]
            backward_rest(out, dmem, dsg, [ql], [torch.full_like(ql, 0.01)])
        else:
            backward_rest(out, dmem, dsg)
        two = grads()
        # N > 1 structure: one stage per decoder layer (last layer first), then the shared nodes and the rest
        out = nugget_step_simul(scorer, inp, use_ilk=True, backward=False) if simul else \
            nugget_step_asr(scorer, inp, 4, backward=False)
        lg = [None] * nl
        for l in reversed(range(nl)):
            lg[l] = attention_backward_layer(out, inp, l)
        if simul:
            ql = out["quantity_loss"]
            backward_rest_layers(out, lg, [ql], [torch.full_like(ql, 0.01)])
        else:
            backward_rest_layers(out, lg)
        three = grads()
        nq = len(leaves) - len(inp.qs)
        for other in (two, three):
            for i, (a, b) in enumerate(zip(one, other)):
                if i >= nq:
                    assert b is None                        # autograd.grad returns dQ instead of accumulating it
                    continue
                assert (a is None) == (b is None)
                if a is not None:
                    assert torch.equal(a, b)


def _simul_full_size(ops, B, T, lengths, tl, use_ilk, seed):
    """A whole simultaneous step at a BASELINE.json size, checked through size-independent properties."""
    from star_amd.nugget import NuggetScorer
    from star_amd.step import make_inputs, nugget_step_simul
    torch.manual_seed(seed)
    d, h, nl = 1024, 8, 4
    L = max(tl)
    scorer = NuggetScorer(d).to(DEV)
    with torch.no_grad():
        scorer[2].bias.fill_(-2.5)
    inp = make_inputs(lengths, T, d, h, L, nl, DEV, seed=seed, target_lengths=tl)
    out = nugget_step_simul(scorer, inp, use_ilk=use_ilk, with_loss=True)
    n = np.asarray(lengths)
    t_np = np.asarray(tl)
    mask = lmask(lengths, T)
    al, rs = out["alphas"].detach().cpu().numpy(), out["resized"].detach().cpu().numpy()
    assert (al[~mask] == 0).all() and (rs[~mask] == 0).all()
    # resize: row sums equal the target lengths (rescale and fix-ups both keep the sum), every weight below 1
    np.testing.assert_allclose(rs.sum(1, dtype=np.float64), t_np, rtol=2e-5)
    assert rs.max() <= 1.0
    ql = ((al.sum(1, dtype=np.float64) - t_np) ** 2).mean()
    assert abs(out["quantity_loss"].item() - ql) <= 1e-4 * ql + 1e-5
    # fire scan: ascending, inside the utterance, last valid frame always fires, count in {L_b, L_b + 1}
    idx, cnt = out["indices"].cpu().numpy(), out["counts"].cpu().numpy()
    assert idx.shape[1] == L + 2 and int(out["max_count"].item()) == cnt.max()
    for b in range(B):
        f = idx[b, :cnt[b]]
        assert (np.diff(f) > 0).all() and f[0] >= 0 and f[-1] == n[b] - 1 and (idx[b, cnt[b]:] == -1).all()
        assert t_np[b] - 1 <= cnt[b] <= t_np[b] + 1
    # three rows of the scan bit-exact against the oracle on the kernel's own resized alphas
    sub = [0, B // 2, B - 1]
    o_mask, o_idx, _ = orc.fire_scan(rs[sub], mask[sub], with_block_mask=False)
    for r, b in enumerate(sub):
        np.testing.assert_array_equal(idx[b, :cnt[b]], o_idx[r, :o_mask[r].sum()])
    # type table / positions / gathers are exact copies
    tab = out["type_table"].cpu().numpy()
    np.testing.assert_array_equal(tab.sum(1), cnt)
    for b in sub:
        assert (tab[b, idx[b, :cnt[b]]] == 1).all()
        j = cnt[b] - 1
        assert torch.equal(out["mem"][b, j], inp.s[b, idx[b, j]])
        assert out["scores_gather"][b, j].item() == out["scores"][b, idx[b, j]].item()
    # gradients: finite; S only receives gradient at fired frames; the scorer gradient is not the quantity part alone
    assert torch.isfinite(out["loss"]).item()
    for p in scorer.parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all() and p.grad.abs().max() > 0
    ds = inp.s.grad.float().abs().sum(-1).cpu().numpy()
    fired = np.zeros((B, T), dtype=bool)
    for b in range(B):
        fired[b, idx[b, :cnt[b]]] = True
    assert (ds[~fired] == 0).all() and (ds[fired] > 0).mean() > 0.99
    assert inp.x.grad is None
    for q in inp.qs:
        assert torch.isfinite(q.grad).all()
    # one layer again with the debug probabilities: masked keys and (ILK) keys j > i + 2 are exact zeros, rows sum
    # to one, and the bias gradient sums to zero over the live keys of a row (softmax is shift invariant)
    K = idx.shape[1]
    kk = out["mem"].detach().view(B, K, h, d // h).permute(0, 2, 1, 3)
    bias = out["scores_gather"].detach().clone().requires_grad_()
    o1, p1 = ops.xattn(inp.qs[0].detach(), kk, kk, bias=bias, key_len=out["counts"], ilk_diag=2 if use_ilk else -1,
                       need_probs=True)
    dead = torch.arange(K, device=DEV)[None, :] >= out["counts"][:, None]
    assert (p1.permute(0, 3, 1, 2)[dead] == 0).all()
    if use_ilk:
        banned = torch.arange(K, device=DEV)[None, :] > torch.arange(L, device=DEV)[:, None] + 2
        assert (p1[:, :, banned] == 0).all()
    dev_ = (p1.sum(-1) - 1).abs()
    assert (dev_ < 1e-4).all(), (dev_.max().item(), (dev_ >= 1e-4).nonzero()[:8].tolist(), out["counts"].tolist())
    o1.backward(torch.randn_like(o1))
    live = (~dead).float()
    tot = (bias.grad * live).sum(1).abs()
    # delta = <O, dO> is formed from the bf16-rounded O, so a row's sum is zero up to 2^-9 of its terms
    assert (tot <= 5e-3 * (bias.grad * live).abs().sum(1) + 1e-6).all() and (bias.grad[dead] == 0).all()
    if use_ilk:
        # query i sees keys j <= i + 2 only: the gradient of the LAST key of a long memory must not depend on G rows
        # of early queries -> zeroing those cotangent rows leaves dS at that key unchanged
        b = 0
        last = idx[b, cnt[b] - 1]
        ref_row = inp.s.grad[b, last].clone()
        gs_saved = [g.clone() for g in inp.gs]
        cut = max(cnt[b] - 4, 0)
        for g in inp.gs:
            g[b, :, :min(cut, g.shape[2])] = 0
        for t in [inp.s] + inp.qs + list(scorer.parameters()):
            t.grad = None
        nugget_step_simul(scorer, inp, use_ilk=True)
        if cut >= 1 and cut <= inp.gs[0].shape[2]:
            assert torch.equal(inp.s.grad[b, last], ref_row)
        for g, s_ in zip(inp.gs, gs_saved):
            g.copy_(s_)
    return out


def test_full_size_simul_config2(ops):
    """BASELINE.json configs[2]: B = 64, MuST-C-shaped log-normal lengths up to 1500 frames, L_b = 0.075 n_b + 2."""
    rng = np.random.default_rng(4321)
    B, T = 64, 1500
    n = np.clip(np.round(50.0 * rng.lognormal(np.log(6.0), 0.6, size=B)), 50, T).astype(int)
    n[0] = T
    tl = np.clip(np.round(0.075 * n) + 2, 4, 512).astype(int)
    _simul_full_size(ops, B, T, [int(v) for v in n], [int(v) for v in tl], False, 11)


def test_full_size_simul_config3(ops):
    """BASELINE.json configs[3]: B = 16, T = 3000, L = 227, infinite-lookback mask on."""
    _simul_full_size(ops, 16, 3000, [3000] * 16, [227] * 16, True, 12)


# ------------------------------------------------------------------------------------------------ f4: streaming
def _chunked_fires(ops, alphas, lengths, chunks, late_length=False):
    """Feed [B,T] weights through star_cif_fire_step in the given chunk sizes; returns per-row lists of positions."""
    B, T = alphas.shape
    st = ops.FireState(B, DEV)
    ln = torch.tensor(lengths, dtype=torch.int32, device=DEV)
    unknown = torch.full_like(ln, 2 ** 31 - 1)
    fires = [[] for _ in range(B)]
    off = 0
    for c in chunks:
        chunk = alphas[:, off:off + c].contiguous()
        # late_length: the end of an utterance only becomes known with the chunk that holds its last frame
        cur_len = torch.where(ln <= off + c, ln, unknown) if late_length else ln
        idx, cnt, _ = ops.cif_fire_step(chunk, cur_len, st, cap=c)
        idx, cnt = idx.cpu().numpy(), cnt.cpu().numpy()
        for b in range(B):
            assert (idx[b, cnt[b]:] == -1).all()
            fires[b].extend(int(v) for v in idx[b, :cnt[b]])
        off += c
    assert off == T and (st.pos.cpu().numpy() == T).all()
    return fires, st


def _chunkings(T, rng):
    out = [[T], [1] * T if T <= 64 else [T // 2, T - T // 2], [16] * (T // 16) + ([T % 16] if T % 16 else [])]
    for _ in range(3):
        sizes, left = [], T
        while left:
            c = int(min(left, rng.integers(1, 200)))
            sizes.append(c)
            left -= c
        out.append(sizes)
    return out


@pytest.mark.parametrize("name", ["cif_plain", "cif_fixup", "cif_ragged"])
def test_fire_step_any_chunking_of_the_goldens(ops, name):
    """f4 first slice: any chunking of the reference's own scan inputs (w2v_simul.py:376-403 goldens) gives the
    reference's fire positions bit for bit, whether the utterance length is known from the start or only at its end."""
    g = load(name)
    lengths = g["lengths"].astype(np.int64)
    alphas = dev(g["resized"].astype(np.float32))
    gold = g["nugget_indices"]
    rng = np.random.default_rng(0)
    for chunks in _chunkings(alphas.shape[1], rng):
        for late in (False, True):
            fires, _ = _chunked_fires(ops, alphas, lengths, chunks, late)
            for b in range(alphas.shape[0]):
                assert fires[b] == [int(v) for v in gold[b] if v >= 0], (name, chunks[:4], late, b)


def test_fire_step_random_vs_whole_scan_and_oracle(ops):
    """Random rows through both recurrences of the kernel: weights in [0, 1] (v_fract chain), rows with weights above
    1, denormal weights and a carry >= 1 at a chunk border (generic chain).  Whole scan == oracle == any chunking, and
    the carried integrator equals the oracle's carry."""
    rng = np.random.default_rng(7)
    B, T = 12, 1000
    a = (rng.random((B, T)) * 0.3).astype(np.float32)
    a[1, ::7] = 0.0
    a[2] *= 6.0                                   # weights up to 1.8: several fires' worth of carry
    a[3, 100:110] = 2.5
    a[4, :50] = 1e-41                             # denormals
    a[5] = 0.0
    a[6, 499] = 1.0
    a[7] = np.float32(1.0)
    lengths = [T, 900, T, T, 333, 10, 500, 64, 1, T, 17, 999]
    mask = lmask(lengths, T)
    o_mask, o_idx, _ = orc.fire_scan(a, mask, with_block_mask=False)
    ad = dev(a)
    ln = torch.tensor(lengths, dtype=torch.int32, device=DEV)
    widx, _, wcnt, _, _ = ops.cif_fire_scan(ad, lengths=ln, cap=T)
    widx, wcnt = widx.cpu().numpy(), wcnt.cpu().numpy()
    for b in range(B):
        np.testing.assert_array_equal(widx[b, :wcnt[b]], o_idx[b, :o_mask[b].sum()])
    # the oracle's carry after the last frame (same recurrence as orc.fire_scan)
    carry = np.zeros(B, dtype=np.float32)
    am = a * mask
    for t in range(T):
        carry = (carry + am[:, t]).astype(np.float32)
        f = (carry >= 1) | (np.asarray(lengths) == t + 1)
        carry = np.where(f, (carry - np.float32(1)).astype(np.float32), carry)
    for chunks in _chunkings(T, rng) + [[499, 1, 500], [3] * 333 + [1]]:
        fires, st = _chunked_fires(ops, ad, lengths, chunks, late_length=True)
        for b in range(B):
            assert fires[b] == [int(v) for v in o_idx[b, :o_mask[b].sum()]], (chunks[:4], b)
        np.testing.assert_array_equal(st.cur.cpu().numpy(), carry)


def test_streaming_segmenter_matches_offline(ops):
    """Incremental scorer + online fire decision + memory append (star_amd.streaming) against the same kernels run on
    the whole utterances: fire positions, memory rows, logits, counts; and the write policy: token i decoded from the
    partial memory as soon as tokens_ready() allows equals row i of the offline infinite-lookback attention."""
    from star_amd.nugget import NuggetScorer
    from star_amd.streaming import StreamingNuggetSegmenter
    torch.manual_seed(3)
    B, T, d, h = 3, 640, 256, 2
    lengths = [640, 500, 130]
    scorer = NuggetScorer(d).to(DEV)
    with torch.no_grad():
        scorer[2].bias.fill_(-2.0)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, T, d, generator=g).to(torch.bfloat16).to(DEV)
    s = torch.randn(B, T, d, generator=g).to(torch.bfloat16).to(DEV)
    ln = torch.tensor(lengths, dtype=torch.int32, device=DEV)
    with torch.no_grad():
        scores = scorer(x).squeeze(-1).float()
    alphas, _ = ops.cif_quantity(scores, torch.zeros_like(ln), lengths=ln)
    idx, nmask, counts, _, sg = ops.cif_fire_scan(alphas, lengths=ln, cap=T, scores=scores)
    kmax = int(counts.max().item())
    mem_off = ops.gather_rows(s, idx[:, :kmax])
    seg = StreamingNuggetSegmenter(scorer, B, d, max_nuggets=kmax + 4, device=DEV)
    q = torch.randn(B, h, 40, d // h, generator=g).to(torch.bfloat16).to(DEV)
    k_off = mem_off.view(B, kmax, h, d // h).permute(0, 2, 1, 3)
    o_off, _ = ops.xattn(q, k_off, k_off, bias=sg[:, :kmax].contiguous(), key_len=counts, ilk_diag=2)
    checked = 0
    off = 0
    for c in [64, 100, 36, 200, 1, 111, 128]:
        known = torch.where(ln <= off + c, ln, torch.full_like(ln, 2 ** 31 - 1))
        seg.push(x[:, off:off + c].contiguous(), s[:, off:off + c].contiguous(), lengths=known)
        off += c
        mem, msc, mpos, mcnt = seg.memory()
        ready = seg.tokens_ready().cpu().numpy()
        m = mcnt.cpu().numpy()
        for b in range(B):
            if off < lengths[b]:
                assert ready[b] == max(m[b] - 2, 0)
            else:
                assert ready[b] == 2 ** 31 - 1
            i = min(int(ready[b]), q.shape[2]) - 1           # the newest token that may be written for row b
            if i < 0:
                continue
            kk = mem[b:b + 1].view(1, -1, h, d // h).permute(0, 2, 1, 3)
            o_i, _ = ops.xattn(q[b:b + 1, :, i:i + 1].contiguous(), kk, kk, bias=msc[b:b + 1].contiguous(),
                               key_len=mcnt[b:b + 1], ilk_diag=seg.ilk_diag_for(i))
            ref = o_off[b, :, i].float()
            assert (o_i[0, :, 0].float() - ref).abs().max() <= 1e-2 * ref.abs().max() + 1e-3
            checked += 1
    assert off == T and checked >= 10
    mem, msc, mpos, mcnt = seg.memory()
    assert torch.equal(mcnt, counts)
    for b in range(B):
        n = int(counts[b])
        assert torch.equal(mpos[b, :n], idx[b, :n])
        assert torch.equal(mem[b, :n], mem_off[b, :n])
        assert torch.equal(msc[b, :n], sg[b, :n])
    assert bool(seg.finished().all())


# ------------------------------------------------------------------------------------------------ ragged scorer
@pytest.mark.parametrize("compact", [True, False])
def test_ragged_scorer_equals_dense_on_valid_frames(ops, compact):
    """star_scorer_fwd_compact runs the GEMM on the gathered valid frames, star_scorer_fwd_ragged (compact=False) on
    the row tiles that hold a valid frame; both write 0 at pad frames: scores at valid frames and every gradient are
    bit-identical to the dense call with the same cotangent (zero at pad frames), for a dense backward
    (quantity-loss-like) and for a backward with an active-row hint (which exercises the frame -> row map of the
    ReLU bits in the compact form)."""
    from star_amd.nugget import NuggetScorer
    torch.manual_seed(4)
    B, T, d = 7, 700, 256
    lengths = [700, 13, 300, 0, 699, 257, 512]
    valid = ops.ValidFrames(lengths, T, DEV, compact=compact)
    mask = torch.from_numpy(lmask(lengths, T)).to(DEV)
    assert valid.n_valid == sum(lengths) and valid.n_live < (B * T + 255) // 256
    assert valid.offsets.tolist() == [0, 700, 713, 1013, 1013, 1712, 1969]
    sc = NuggetScorer(d).to(DEV)
    x = torch.randn(B, T, d, device=DEV).to(torch.bfloat16)
    g = torch.randn(B, T, device=DEV) * mask
    res = []
    for v in (None, valid):
        for hinted in (False, True):
            xr = x.clone().requires_grad_()
            sc.zero_grad()
            s = sc(xr, v).squeeze(-1)
            gg = g.clone()
            if hinted:
                keep = torch.zeros_like(mask)
                keep[:, ::5] = True
                gg = gg * keep
                # the row list may name pad frames too (the top-k select lists its pad slots): their score gradient is
                # zero, and in the compact form their ReLU bits are looked up at row 0
                rows = keep.view(-1).nonzero().view(-1).to(torch.int32)
                sc.set_active_rows(rows)
            s.backward(gg)
            res.append((s.detach().clone(), xr.grad.clone(), [p.grad.clone() for p in sc.parameters()]))
    for exact, (a, b) in ((False, (res[0], res[2])), (True, (res[1], res[3]))):
        assert torch.equal(a[0][mask], b[0][mask]) and (b[0][~mask] == 0).all()
        assert torch.equal(a[1], b[1])
        for pa, pb in zip(a[2], b[2]):
            if exact:                      # same row list -> same kernels on the same operands
                assert torch.equal(pa, pb)
            else:                          # all B*T rows against the valid rows: other tiles, other summation order
                close(pb, pa, "scorer parameter gradient", 1e-5)
    with pytest.raises(ValueError):
        sc(x[:, :100], valid)
    # a batch without a single valid frame: nothing is computed, every score is 0
    empty = ops.ValidFrames([0, 0], 300, DEV, compact=compact)
    with torch.no_grad():
        assert (sc(x[:2, :300].contiguous(), empty) == 0).all()
    xe = x[:2, :300].contiguous().requires_grad_()
    sc.zero_grad()
    sc(xe, empty).sum().backward()
    assert (xe.grad == 0).all() and all((p.grad == 0).all() for p in sc.parameters())


# ------------------------------------------------------------------------------------------------ f-3, second half
@pytest.mark.parametrize("name", ["gatherln_a", "gatherln_b"])
def test_gather_layernorm_golden_gpu(ops, name):
    """ln_f(hidden)[idx] computed as ln_f(hidden[idx]) against the reference call site (fp32: rtol 1e-5; the row
    statistics are two-pass fp32, torch's are Welford)."""
    g = load(name)
    h = dev(g["hidden"]).requires_grad_()
    w = dev(g["weight"]).requires_grad_()
    b = dev(g["bias"]).requires_grad_()
    out = ops.gather_layernorm(h, dev(g["idx"]), w, b, float(g["eps"]))
    np.testing.assert_allclose(out.detach().cpu().numpy(), g["out"], rtol=1e-5, atol=2e-6)
    (out * dev(g["gout"])).sum().backward()
    np.testing.assert_allclose(h.grad.cpu().numpy(), g["dhidden"], rtol=1e-4, atol=5e-6)
    np.testing.assert_allclose(w.grad.cpu().numpy(), g["dweight"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(b.grad.cpu().numpy(), g["dbias"], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("B,T,K,d,dtype", [(32, 1500, 375, 1024, torch.bfloat16), (3, 70, 90, 768, torch.float32),
                                            (2, 33, 5, 8, torch.float32), (4, 129, 40, 1536, torch.bfloat16)])
def test_gather_layernorm_random_vs_oracle(ops, B, T, K, d, dtype):
    """Random rows incl. -1 pads and duplicate indices (their gradients add up in d hidden), K > T, the configs[1] size;
    bf16 inputs: the oracle runs in fp32 on the bf16-rounded inputs, outputs compared at bf16 resolution."""
    from star_amd.nugget import gather_nuggets_layernorm
    rng = np.random.default_rng(B * 1000 + d)
    idx = np.sort(rng.integers(0, T, size=(B, K)), axis=1)
    idx[:, -max(K // 8, 1):] = -1
    h32 = (torch.randn(B, T, d) * 2 + 0.5).to(dtype).float()
    ln = torch.nn.LayerNorm(d, eps=1e-5)
    with torch.no_grad():
        ln.weight.copy_(1 + 0.3 * torch.randn(d))
        ln.bias.copy_(0.2 * torch.randn(d))
    gout32 = torch.randn(B, K, d).to(dtype).float()
    hr = h32.clone().requires_grad_()
    wr, br = ln.weight.detach().clone().requires_grad_(), ln.bias.detach().clone().requires_grad_()
    ref = orc.gather_layernorm(hr, torch.from_numpy(idx), wr, br, 1e-5)
    (ref * gout32).sum().backward()
    lnd = torch.nn.LayerNorm(d, eps=1e-5).to(DEV)
    lnd.load_state_dict(ln.state_dict())
    hd = h32.to(dtype).to(DEV).requires_grad_()
    out = gather_nuggets_layernorm(hd, dev(idx), lnd)
    assert out.dtype == dtype and out.shape == (B, K, d)
    (out.float() * gout32.to(DEV)).sum().backward()
    tol = 1e-2 if dtype == torch.bfloat16 else 1e-5
    close(out, ref, "out", tol)
    close(hd.grad, hr.grad, "d hidden", 2 * tol)
    close(lnd.weight.grad, wr.grad, "d weight", 2 * tol if dtype == torch.bfloat16 else 1e-4)
    close(lnd.bias.grad, br.grad, "d bias", 2 * tol if dtype == torch.bfloat16 else 1e-4)
    never = np.ones((B, T), dtype=bool)
    for bb in range(B):
        never[bb, np.where(idx[bb] < 0, T - 1, idx[bb])] = False
    assert (hd.grad.float().cpu().numpy()[never] == 0).all()          # frames that were not gathered: exact zeros


def test_fire_scan_many_utterances_and_small_cap(ops):
    """Huge batches put several utterances into one workgroup of the scan (one serial lane each, both recurrences side
    by side in one wave); a cap below the fire count drops the late fires but still counts them."""
    rng = np.random.default_rng(11)
    B, T = 4200, 48
    a = (rng.random((B, T)) * 0.6).astype(np.float32)
    a[::7] *= 3.0                                   # every seventh row takes the generic chain (weights above 1)
    lengths = rng.integers(1, T + 1, size=B)
    mask = lmask(lengths, T)
    o_mask, o_idx, _ = orc.fire_scan(a, mask, with_block_mask=False)
    ln = torch.from_numpy(lengths.astype(np.int32)).to(DEV)
    idx, m, cnt, mx, _ = ops.cif_fire_scan(dev(a), lengths=ln, cap=T)
    idx, cnt = idx.cpu().numpy(), cnt.cpu().numpy()
    np.testing.assert_array_equal(cnt, o_mask.sum(1))
    kf = o_idx.shape[1]
    np.testing.assert_array_equal(idx[:, :kf], o_idx)
    assert int(mx.item()) == kf
    # the same rows in two chunks through the streaming step
    st = ops.FireState(B, DEV)
    i1, c1, _ = ops.cif_fire_step(dev(a[:, :20].copy()), ln, st, cap=20)
    i2, c2, _ = ops.cif_fire_step(dev(a[:, 20:].copy()), ln, st, cap=T - 20)
    np.testing.assert_array_equal((c1 + c2).cpu().numpy(), cnt)
    both = torch.cat([i1, i2], 1).cpu().numpy()
    for b in range(0, B, 97):
        got = both[b][both[b] >= 0]
        np.testing.assert_array_equal(got, o_idx[b][o_idx[b] >= 0])
    # cap smaller than the number of fires
    cap = 3
    idx3, m3, cnt3, _, _ = ops.cif_fire_scan(dev(a), lengths=ln, cap=cap)
    np.testing.assert_array_equal(cnt3.cpu().numpy(), cnt)
    np.testing.assert_array_equal(idx3.cpu().numpy()[:, :min(cap, kf)], o_idx[:, :cap])


# ------------------------------------------------------------------------------------------------ fused segmenter
@pytest.mark.parametrize("B,T,frac", [(64, 1500, 0.26), (16, 3000, 1.0), (5, 333, 0.7), (3, 17, 1.0),
                                      (4, 1024, 1.0), (4, 1025, 0.5), (3, 2049, 0.8), (2, 1, 1.0)])
def test_cif_segment_equals_the_chain_of_kernels(ops, B, T, frac):
    """star_cif_segment against cif_quantity -> cif_resize -> cif_fire_scan -> gather_rows -> type_table: every output
    bit-identical, and the score gradient (quantity loss + gathered-score cotangent) equal to autograd's sum of the two."""
    rng = np.random.default_rng(B + T)
    lengths = np.maximum((rng.random(B) * frac * T).astype(np.int64), 1)
    lengths[0] = T
    tl = np.clip(np.round(0.075 * lengths) + 2, 1, None).astype(np.int32)
    scores = (torch.randn(B, T) - 2.5).to(DEV)
    scores[1 % B, : min(5, T)] = 6.0                      # weights near 1 after resize: fix-up rounds
    ln, tld = dev(lengths.astype(np.int32)), dev(tl)
    cap = int(tl.max()) + 2
    s1 = scores.clone().requires_grad_()
    a1, q1 = ops.cif_quantity(s1, tld, lengths=ln)
    r1 = ops.cif_resize(a1, tld)
    i1, m1, c1, x1, _ = ops.cif_fire_scan(r1, lengths=ln, cap=cap)
    g1 = ops.gather_rows(s1, i1)
    t1, p1 = ops.type_table(i1, m1, T)
    s2 = scores.clone().requires_grad_()
    a2, q2, r2, i2, m2, c2, x2, g2, t2, p2 = ops.cif_segment(s2, tld, ln, cap=cap)
    for u, v in ((a1, a2), (q1, q2), (r1, r2), (i1, i2), (m1, m2), (c1, c2), (x1, x2), (g1, g2), (t1, t2), (p1, p2)):
        assert torch.equal(u.detach(), v.detach())
    gg = torch.randn(B, cap, device=DEV) * m1
    (0.01 * q1 + (g1 * gg).sum()).backward()
    (0.01 * q2 + (g2 * gg).sum()).backward()
    np.testing.assert_allclose(s2.grad.cpu().numpy(), s1.grad.cpu().numpy(), rtol=1e-6, atol=1e-9)
    # either gradient source alone
    for use_q, use_g in ((True, False), (False, True)):
        s3 = scores.clone().requires_grad_()
        out = ops.cif_segment(s3, tld, ln, cap=cap)
        s4 = scores.clone().requires_grad_()
        a4, q4 = ops.cif_quantity(s4, tld, lengths=ln)
        g4 = ops.gather_rows(s4, i1)
        ((out[1] if use_q else 0) + ((out[7] * gg).sum() if use_g else 0)).backward()
        ((q4 if use_q else 0) + ((g4 * gg).sum() if use_g else 0)).backward()
        assert torch.equal(s3.grad, s4.grad)


@pytest.mark.parametrize("name", ["cif_plain", "cif_fixup", "cif_ragged"])
def test_cif_segment_golden(ops, name):
    """The fused segmenter on the reference's own vectors (scores, lengths, target lengths of the cif goldens)."""
    g = load(name)
    T = g["scores"].shape[1]
    out = ops.cif_segment(dev(g["scores"].astype(np.float32)), dev(g["target_lengths"].astype(np.int32)),
                          dev(g["lengths"].astype(np.int32)), cap=T)
    alphas, qloss, resized, idx, nmask = out[:5]
    np.testing.assert_allclose(alphas.cpu().numpy(), g["alphas"], rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(qloss.item(), float(g["quantity_loss"]), rtol=1e-5)
    kf = g["nugget_indices"].shape[1]
    if np.array_equal(resized.cpu().numpy(), g["resized"]):           # same resized bits -> same fires, exactly
        np.testing.assert_array_equal(idx[:, :kf].cpu().numpy(), g["nugget_indices"])
        np.testing.assert_array_equal(nmask[:, :kf].cpu().numpy(), g["nugget_mask"].astype(bool))
    else:
        np.testing.assert_allclose(resized.cpu().numpy(), g["resized"], rtol=2e-5, atol=1e-7)
    o_mask, o_idx, _ = orc.fire_scan(resized.cpu().numpy(), lmask(g["lengths"], T), with_block_mask=False)
    np.testing.assert_array_equal(idx[:, :o_idx.shape[1]].cpu().numpy(), o_idx)


# ------------------------------------------------------------------------------------------------ dK/dV for L <= 256
@pytest.mark.parametrize("B,h,L,K,hd,ilk,p", [(16, 8, 227, 229, 128, 2, 0.0), (3, 4, 256, 300, 128, -1, 0.0),
                                              (2, 8, 129, 600, 64, -1, 0.0), (2, 2, 200, 90, 128, 2, 0.2)])
def test_dkv_small_256_rows_vs_general_kernel(ops, B, h, L, K, hd, ilk, p):
    """Targets of 129..256 tokens take the barrier-free dK/dV kernel with a 256-row Q / dO block in LDS (configs[3]:
    L = 227): same results as the general dK/dV kernel (forced through the test override) up to bf16 rounding of the
    summation order, and both against the fp32 oracle."""
    g = torch.Generator().manual_seed(L + K)
    q = torch.randn(B, h, L, hd, generator=g).to(torch.bfloat16).to(DEV).requires_grad_()
    k = torch.randn(B, h, K, hd, generator=g).to(torch.bfloat16).to(DEV).requires_grad_()
    v = torch.randn(B, h, K, hd, generator=g).to(torch.bfloat16).to(DEV).requires_grad_()
    bias = torch.randn(B, K, generator=g).to(DEV).requires_grad_()
    kl = torch.randint(max(K // 2, 1), K + 1, (B,), generator=g).to(torch.int32).to(DEV)
    kl[0] = K
    go = torch.randn(B, h, L, hd, generator=g).to(torch.bfloat16).to(DEV)
    res = []
    for mode in (0, 1):
        ops.test_override(ops.OVR_XA_DKV, mode)
        try:
            ops.reset_dropout_state(77)
            o, _ = ops.xattn(q, k, v, bias=bias, key_len=kl, ilk_diag=ilk, dropout_p=p)
            res.append(torch.autograd.grad(o, (q, k, v, bias), go))
        finally:
            ops.test_override(ops.OVR_XA_DKV, 0)
    for a, b, what in zip(res[0], res[1], ("dQ", "dK", "dV", "dbias")):
        close(a, b, what, 1e-2)
    if p == 0.0:
        qr, kr, vr = (t.detach().float().cpu().requires_grad_() for t in (q, k, v))
        br = bias.detach().cpu().requires_grad_()
        km = torch.arange(K)[None, :] < kl.cpu()[:, None]
        oref, _ = orc.xattn_core(qr, kr, vr, km, br, ilk >= 0)
        (oref * go.float().cpu()).sum().backward()
        for a, b, what in zip(res[0], (qr.grad, kr.grad, vr.grad, br.grad), ("dQ", "dK", "dV", "dbias")):
            close(a, b, what + " vs oracle", 2e-2)


# ------------------------------------------------------------------------------------------------ 8f item 1
@pytest.mark.parametrize("ilk", [False, True])
def test_share_encoder_kv_stack_vs_oracle(ops, ilk):
    """the c_attn of all decoder layers as one projection of the encoder memory
    (my_transformer.py:500-508, 917-928).  A stack of nl cross-attention modules fed from share_encoder_kv() against
    the per-layer fp32 restatement with the oracle's core: outputs, d memory, d scorer logits and every parameter
    gradient; the attention backwards must have written dK / dV in place (no copies, no per-layer span allocations)."""
    from star_amd.nugget import NuggetCrossAttention, share_encoder_kv, share_scorer_logits
    torch.manual_seed(21)
    d, h, B, L, K, nl = 256, 4, 3, 20, 37, 3
    mods = [NuggetCrossAttention(d, h, layer_idx=i).to(DEV, torch.bfloat16) for i in range(nl)]
    with torch.no_grad():
        for m in mods:
            for prm in m.parameters():
                prm.copy_(torch.randn_like(prm) * (0.05 if prm.dim() == 2 else 0.02))
    xs = [torch.randn(B, L, d, device=DEV).to(torch.bfloat16).requires_grad_() for _ in range(nl)]
    enc = torch.randn(B, K, d, device=DEV).to(torch.bfloat16).requires_grad_()
    sc = torch.randn(B, K, device=DEV).requires_grad_()
    klen = [K, K - 9, 5]
    kmask = lmask(klen, K)
    add = dev(((1.0 - kmask[:, None, None, :].astype(np.float32)) * np.finfo(np.float32).min))
    gouts = [torch.randn(B, L, d, device=DEV).to(torch.bfloat16) for _ in range(nl)]

    kvs = share_encoder_kv(mods, enc)
    logits = share_scorer_logits(sc, nl, h, attention_mask=add)
    outs = [m(x, encoder_kv=kv, scorer_logits=lg, limit_cross_attn=ilk)[0]
            for m, x, kv, lg in zip(mods, xs, kvs, logits)]
    torch.autograd.backward(outs, gouts)
    stats = kvs[0].grad_slot.buffer.stats
    assert stats == {"in_place": 2 * nl, "copied": 0, "zeroed": 0}, stats
    assert kvs[0].grad_slot.buffer.buf is None                      # released by the projection's backward

    er, sr = (t.detach().float().cpu().requires_grad_() for t in (enc, sc))
    split = lambda t: t.view(t.shape[0], t.shape[1], h, d // h).permute(0, 2, 1, 3)
    ys, Ps, xrs = [], [], []
    for m, x in zip(mods, xs):
        P = {n: t.detach().float().cpu().requires_grad_() for n, t in m.named_parameters()}
        xr = x.detach().float().cpu().requires_grad_()
        q = split(xr @ P["q_attn.weight"] + P["q_attn.bias"])
        kv = er @ P["c_attn.weight"] + P["c_attn.bias"]
        o_ref, _ = orc.xattn_core(q, split(kv[..., :d]), split(kv[..., d:]), torch.from_numpy(kmask), sr, ilk)
        ys.append(o_ref.permute(0, 2, 1, 3).reshape(B, L, d) @ P["c_proj.weight"] + P["c_proj.bias"])
        Ps.append(P)
        xrs.append(xr)
    sum((y * g.float().cpu()).sum() for y, g in zip(ys, gouts)).backward()

    for i in range(nl):
        close(outs[i], ys[i], f"layer {i} output", 3e-2)
        close(xs[i].grad, xrs[i].grad, f"layer {i} d hidden", 3e-2)
        for n, t in mods[i].named_parameters():
            close(t.grad, Ps[i][n].grad, f"layer {i} d {n}", 3e-2)
    close(enc.grad, er.grad, "d encoder memory", 3e-2)
    close(sc.grad, sr.grad, "d scorer_logits", 3e-2)


def test_share_encoder_kv_same_kv_used_twice_and_generation_cache(ops):
    """The same layer's K / V feeding two attention calls: the second backward must not overwrite the first one's
    dK / dV in the shared buffer (autograd adds the two); under no_grad the list is the K/V cache of all layers and
    gives the per-layer modules' outputs token after token."""
    from star_amd.nugget import NuggetCrossAttention, share_encoder_kv
    torch.manual_seed(22)
    d, h, B, L, K = 256, 4, 2, 6, 33
    m = NuggetCrossAttention(d, h).to(DEV, torch.bfloat16)
    enc = torch.randn(B, K, d, device=DEV).to(torch.bfloat16).requires_grad_()
    x1, x2 = (torch.randn(B, L, d, device=DEV).to(torch.bfloat16) for _ in range(2))
    g1, g2 = (torch.randn(B, L, d, device=DEV).to(torch.bfloat16) for _ in range(2))

    def grads(shared):
        m.zero_grad()
        enc.grad = None
        if shared:
            kv = share_encoder_kv([m], enc)[0]
            o1, o2 = m(x1, encoder_kv=kv)[0], m(x2, encoder_kv=kv)[0]
        else:
            o1, o2 = m(x1, encoder_hidden_states=enc)[0], m(x2, encoder_hidden_states=enc)[0]
        torch.autograd.backward([o1, o2], [g1, g2])
        return o1.detach(), o2.detach(), enc.grad.clone(), m.c_attn.weight.grad.clone()

    ref, got = grads(False), grads(True)
    for a, b, what in zip(got, ref, ("o1", "o2", "d memory", "d c_attn.weight")):
        close(a, b, what, 2e-2)

    mods = [NuggetCrossAttention(d, h, layer_idx=i).to(DEV, torch.bfloat16).eval() for i in range(2)]
    with torch.no_grad():
        cache = share_encoder_kv(mods, enc)
        for _ in range(3):                                           # three decoded tokens, one projection
            tok = torch.randn(B, 1, d, device=DEV).to(torch.bfloat16)
            for mm, kv in zip(mods, cache):
                close(mm(tok, encoder_kv=kv)[0], mm(tok, encoder_hidden_states=enc)[0], "cached token", 1e-2)


# ------------------------------------------------------------------------------------------------ TN dW1 GEMM
@pytest.mark.parametrize("B,T,d,every", [(3, 500, 384, 1), (2, 1100, 1024, 3), (1, 77, 128, 1), (5, 640, 256, 7),
                                         (12, 1500, 256, 1)])
def test_scorer_backward_tn_gemm_equals_nt_gemm(ops, B, T, d, every):
    """The scorer backward forms U = G'^T X_sel with the TN kernel straight from the row-major G' and the gathered rows
    of x; the NT kernel on transposed copies (forced through star_test_override) is the reference: same operands, same
    split and K-tile order -> the gradients agree to fp32 rounding (and are compared with the oracle elsewhere)."""
    from star_amd.nugget import NuggetScorer
    torch.manual_seed(B * 1000 + d)
    sc = NuggetScorer(d).to(DEV)
    x = torch.randn(B, T, d, device=DEV).to(torch.bfloat16)
    keep = torch.zeros(B, T, dtype=torch.bool, device=DEV)
    keep[:, ::every] = True
    g = torch.randn(B, T, device=DEV) * keep
    rows = keep.view(-1).nonzero().view(-1).to(torch.int32)
    res = []
    for form in (2, 1):                                            # 2 = TN (the default), 1 = NT
        prev = ops.test_override(ops.OVR_SCORER_NT, form)
        try:
            xr = x.clone().requires_grad_()
            sc.zero_grad()
            s = sc(xr).squeeze(-1)
            if every > 1:
                sc.set_active_rows(rows)
            s.backward(g)
            res.append((xr.grad.clone(), [p.grad.clone() for p in sc.parameters()]))
        finally:
            ops.test_override(ops.OVR_SCORER_NT, prev)
    assert torch.equal(res[0][0], res[1][0])                      # dX does not go through the dW1 GEMM
    for a, b in zip(res[0][1], res[1][1]):
        close(a, b, "scorer parameter gradient, TN vs NT", 1e-6)


# ------------------------------------------------------------------------------------------------ graph capture
@pytest.mark.parametrize("form,every", [(1, 4), (2, 1), (2, 4)])
def test_scorer_backward_under_graph_capture_equals_eager(ops, form, every):
    """Captured in a HIP graph, star_scorer_bwd forks onto the auxiliary stream (X_sel^T, the zero-fill of dx and the
    dX GEMM beside the G' pass / dW1 GEMM / reductions); eager launches use one stream.  Same kernels on the same
    operands: every gradient must come out bit-identical, and replaying the graph must reproduce them (dx is re-zeroed
    and re-written by every replay).  form: 1 = NT dW1 GEMM, 2 = TN."""
    from star_amd.nugget import NuggetScorer
    torch.manual_seed(31)
    B, T, d = 4, 600, 256
    sc = NuggetScorer(d).to(DEV)
    x = torch.randn(B, T, d, device=DEV).to(torch.bfloat16).requires_grad_()
    keep = torch.zeros(B, T, dtype=torch.bool, device=DEV)
    keep[:, ::every] = True
    g = torch.randn(B, T, device=DEV) * keep
    rows = keep.view(-1).nonzero().view(-1).to(torch.int32)
    params = list(sc.parameters())
    prev = ops.test_override(ops.OVR_SCORER_NT, form)
    stream = torch.cuda.Stream()
    try:
        with torch.cuda.stream(stream):
            def step():
                s = sc(x).squeeze(-1)
                sc.set_active_rows(rows)
                return torch.autograd.grad(s, [x] + params, g)
            eager = [t.clone() for t in step()]
            for _ in range(2):
                step()
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=stream):
                outs = step()
            for _ in range(3):
                for o in outs:
                    o.fill_(float("nan"))
                graph.replay()
                torch.cuda.synchronize()
                for a, b in zip(outs, eager):
                    assert torch.equal(a, b)
    finally:
        ops.test_override(ops.OVR_SCORER_NT, prev)


def test_scorer_backward_tn_row_range_limit(ops):
    """The TN dW1 GEMM keeps the row ids of a workgroup's r range in LDS (7168 ids): R = 114 688 rows at 16 splits is the
    largest case it takes, one more 256-row block goes to the NT form.  Both sides of the limit against the forced NT
    form on the same rows."""
    from star_amd.nugget import NuggetScorer
    torch.manual_seed(41)
    B, T, d = 77, 1500, 128
    sc = NuggetScorer(d).to(DEV)
    x = torch.randn(B, T, d, device=DEV).to(torch.bfloat16)
    g_full = torch.randn(B, T, device=DEV)
    for R in (114688, 114688 + 256):
        rows = torch.arange(R, dtype=torch.int32, device=DEV)
        g = g_full.clone().view(-1)
        g[R:] = 0
        g = g.view(B, T)
        res = []
        for form in (0, 1):                                            # 0 = the library's choice, 1 = NT
            prev = ops.test_override(ops.OVR_SCORER_NT, form)
            try:
                sc.zero_grad()
                s = sc(x).squeeze(-1)
                sc.set_active_rows(rows)
                s.backward(g)
                res.append([p.grad.clone() for p in sc.parameters()])
            finally:
                ops.test_override(ops.OVR_SCORER_NT, prev)
        for a, b in zip(*res):
            close(a, b, f"scorer parameter gradient at R = {R}", 1e-6)


# ------------------------------------------------------------------------------------------------ gather backward paths
@pytest.mark.parametrize("B,T,K,d,npad,dtype", [(3, 1500, 116, 1024, 90, torch.bfloat16),     # configs[2]-shaped pads
                                                (2, 300, 40, 1024, 7, torch.float32),       # fp32: 16-byte vectors of 4
                                                (2, 200, 33, 2048, 20, torch.bfloat16),     # 64 vectors per wave quarter
                                                (2, 130, 20, 4096, 9, torch.bfloat16),      # rows too wide for the fast paths
                                                (2, 90, 17, 100, 5, torch.bfloat16),        # 8-byte vectors
                                                (4, 64, 64, 128, 0, torch.bfloat16)])       # every frame of a tile taken once
def test_gather_backward_paths_bit_exact(ops, B, T, K, d, npad, dtype):
    """Row-gather backward: single-source rows (listed, loads in flight across the zero stores), frames with many
    sources (summed by the whole workgroup) and untouched frames, each path and vector width: equal, bit for bit, to
    the fp32 sum in ascending slot order rounded once (the documented rule)."""
    g = torch.Generator().manual_seed(B * T + K + d)
    idx = torch.stack([torch.randperm(T, generator=g)[:K].sort().values for _ in range(B)])
    if npad:
        idx[:, K - npad:] = -1                       # pad slots: all of them address frame T - 1
        idx[0, 1] = idx[0, 0]                        # and one ordinary duplicate
    src = torch.zeros(B, T, d, dtype=dtype, device=DEV, requires_grad=True)
    out = ops.gather_rows(src, idx.to(DEV))
    gout = torch.randn(B, K, d, generator=g).to(dtype)
    out.backward(gout.to(DEV))
    acc = torch.zeros(B, T, d, dtype=torch.float32)
    tgt = torch.where(idx < 0, idx + T, idx)
    for b in range(B):
        for j in range(K):                           # ascending slot order
            acc[b, tgt[b, j]] += gout[b, j].float()
    assert torch.equal(src.grad.cpu(), acc.to(dtype))
