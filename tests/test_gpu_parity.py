"""GPU parity: the HIP kernels (through the C ABI) against the CPU oracle and the reference golden vectors.

Bars: indices / masks / fire positions / type tables bit-exact; row gather bit-exact;
fp32 scalar paths rtol 1e-5; bf16-MFMA paths against the fp32 oracle evaluated on the same bf16-rounded inputs
with the tolerances written at each assert.
"""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import nugget_oracle as orc
from tests.test_oracle_golden import assert_select_equal_up_to_ties, lmask, load, select_keys

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
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


# ------------------------------------------------------------------------------------------------ select
SELECT = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(G, "select_*.npz")) if "counts" not in p)


@pytest.mark.parametrize("name", SELECT)
@pytest.mark.parametrize("use_mask", [True, False])
def test_select_golden(ops, name, use_mask):
    g = load(name)
    B, T = g["scores"].shape
    mask = lmask(g["lengths"], T)
    kw = dict(attn_mask=dev(mask)) if use_mask else dict(lengths=dev(g["lengths"], torch.int32))
    nmask, idx, sg, table, pos = ops.topk_select(dev(g["scores"]), rate=float(g["rate"]), **kw)
    nmask, idx, sg, table, pos = (t.cpu().numpy() for t in (nmask, idx, sg, table, pos))
    # the oracle uses the same tie rule as the kernel -> exact equality
    o_mask, o_idx, o_sg = orc.topk_select(g["scores"], mask, float(g["rate"]))
    o_table, o_pos = orc.type_table(T, o_idx, o_mask)
    np.testing.assert_array_equal(nmask, o_mask)
    np.testing.assert_array_equal(idx, o_idx)
    np.testing.assert_array_equal(sg, o_sg)
    np.testing.assert_array_equal(table, o_table)
    np.testing.assert_array_equal(pos, o_pos)
    # and equality with the reference wherever its answer does not hinge on its unstable sort
    np.testing.assert_array_equal(nmask, g["nugget_mask"])
    assert_select_equal_up_to_ties(idx, g["indices"], nmask, select_keys(g["scores"], g["lengths"]), sg,
                                   g["scores_gather"])


@pytest.mark.parametrize("name", ["selectn_a", "selectn_b", "selectn_over"])
def test_select_given_counts_golden(ops, name):
    """apply_nugget_compression(..., n_nugget=) (w2v_asr.py:381,452: counts from the word delimiters, no clamp to the
    valid frames): kernel = oracle bit for bit, = the reference wherever its answer does not hinge on its unstable sort,
    through the drop-in method itself."""
    from star_amd.nugget import NuggetOpsMixin
    g = load(name)
    B, T = g["scores"].shape
    mask = lmask(g["lengths"], T)
    me = NuggetOpsMixin()
    me.nugget_compress_rate = 4
    nmask, idx, sg = me.apply_nugget_compression(torch.zeros(B, T, 2, device=DEV), dev(mask), nugget_scores=dev(g["scores"]),
                                                 n_nugget=dev(g["n_nugget"]))
    nmask, idx, sg = (t.cpu().numpy() for t in (nmask, idx, sg))
    o_mask, o_idx, o_sg = orc.topk_select(g["scores"], mask, n_nugget=g["n_nugget"])
    np.testing.assert_array_equal(nmask, o_mask)
    np.testing.assert_array_equal(idx, o_idx)
    np.testing.assert_array_equal(sg, o_sg)
    np.testing.assert_array_equal(nmask, g["nugget_mask"])
    assert_select_equal_up_to_ties(idx, g["indices"], nmask, select_keys(g["scores"], g["lengths"]), sg,
                                   g["scores_gather"])


@pytest.mark.parametrize("B,T,rate", [(8, 1500, 4), (8, 1500, 2), (4, 1500, 12), (3, 3000, 8), (5, 257, 4), (2, 8191, 4),
                                      # around the 1024 threads of a workgroup
                                      (3, 1023, 4), (3, 1024, 2), (2, 1025, 8), (2, 2049, 4), (4, 5, 1)])
def test_select_random_vs_oracle(ops, B, T, rate):
    rng = np.random.default_rng(B * 1000 + T + rate)
    scores = rng.standard_normal((B, T)).astype(np.float32)
    scores[0, ::7] = scores[0, 3]          # many exact ties
    lengths = rng.integers(1, T + 1, size=B)
    lengths[0] = T
    mask = lmask(lengths, T)
    nmask, idx, sg, table, pos, rows = ops.topk_select(dev(scores), lengths=dev(lengths, torch.int32), rate=rate,
                                                       want_rows=True)
    flat = idx.cpu().numpy() + np.arange(B)[:, None] * T
    assert rows.dtype == torch.int32 and np.array_equal(rows.cpu().numpy(), flat.reshape(-1))
    o_mask, o_idx, o_sg = orc.topk_select(scores, mask, rate)
    o_table, o_pos = orc.type_table(T, o_idx, o_mask)
    np.testing.assert_array_equal(nmask.cpu().numpy(), o_mask)
    np.testing.assert_array_equal(idx.cpu().numpy(), o_idx)
    np.testing.assert_array_equal(sg.cpu().numpy(), o_sg)
    np.testing.assert_array_equal(table.cpu().numpy(), o_table)
    np.testing.assert_array_equal(pos.cpu().numpy(), o_pos)
    # stand-alone type table entry point
    t2, p2 = ops.type_table(idx, nmask, T)
    np.testing.assert_array_equal(t2.cpu().numpy(), o_table)
    np.testing.assert_array_equal(p2.cpu().numpy(), o_pos)


def test_select_scores_gather_backward(ops):
    scores = torch.randn(3, 50, device=DEV, requires_grad=True)
    nmask, idx, sg = ops.topk_select(scores, lengths=dev(np.array([50, 20, 7]), torch.int32), rate=4, want_table=False)
    w = torch.randn_like(sg)
    (sg * w).sum().backward()
    ref = torch.zeros_like(scores)
    ref.scatter_add_(1, idx, w)
    assert torch.equal(scores.grad, ref)


# ------------------------------------------------------------------------------------------------ gather
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("B,T,K,d", [(3, 70, 9, 768), (2, 129, 40, 1024), (2, 33, 5, 96)])
def test_gather_rows(ops, dtype, B, T, K, d):
    g = torch.Generator().manual_seed(B * T + d)
    src = torch.randn(B, T, d, generator=g).to(dtype)
    idx = torch.stack([torch.randperm(T, generator=g)[:K].sort().values for _ in range(B)])
    idx[0, -2:] = -1                         # simul-style pads (both address frame T-1)
    idx[1, -1] = idx[1, 0]                   # a genuine duplicate
    s_dev = src.to(DEV).requires_grad_()
    out = ops.gather_rows(s_dev, idx.to(DEV))
    ref_src = src.clone().requires_grad_()
    ref = orc.gather_rows(ref_src, idx)
    assert torch.equal(out.cpu(), ref)                      # bit-exact copy
    gout = torch.randn(B, K, d, generator=g).to(dtype)
    out.backward(gout.to(DEV))
    # oracle accumulates in fp32 then rounds once (the kernel's rule); build it explicitly
    acc = torch.zeros(B, T, d, dtype=torch.float32)
    for b in range(B):
        for j in range(K):
            acc[b, idx[b, j]] += gout[b, j].float()
    assert torch.equal(s_dev.grad.cpu(), acc.to(dtype))


def test_gather_scores_negative_index(ops):
    scores = torch.randn(4, 31, device=DEV, requires_grad=True)
    idx = torch.tensor([[0, 5, 30, -1], [3, -1, -1, -1], [1, 2, 3, 4], [30, -1, -1, -1]], device=DEV)
    out = ops.gather_rows(scores, idx)
    ref = scores.detach().cpu()[torch.arange(4).unsqueeze(1), idx.cpu()]
    assert torch.equal(out.detach().cpu(), ref)
    w = torch.randn(4, 4, device=DEV)
    (out * w).sum().backward()
    acc = torch.zeros(4, 31)
    for b in range(4):
        for j in range(4):
            acc[b, idx[b, j].item()] += w[b, j].item()
    np.testing.assert_allclose(scores.grad.cpu().numpy(), acc.numpy(), rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("B,T,K,dtype", [(32, 1500, 375, torch.float32), (5, 700, 900, torch.float32),
                                         (3, 6144, 50, torch.float32), (2, 7000, 40, torch.float32)])
def test_gather_scores_backward_random_duplicates(ops, B, T, K, dtype):
    """Score gather backward (d = 1) with many duplicates and -1 pads: equal to an fp32 scatter-add in slot order
    (the LDS-resident variant up to T, K = 6144; the tiled kernel beyond)."""
    g = torch.Generator().manual_seed(T + K)
    scores = torch.randn(B, T, generator=g).to(dtype).to(DEV).requires_grad_()
    idx = torch.randint(-1, T, (B, K), generator=g)
    idx[:, K // 2:] = torch.where(torch.rand(B, K - K // 2, generator=g) < 0.5, torch.tensor(-1), idx[:, K // 2:])
    w = torch.randn(B, K, generator=g).to(dtype)
    out = ops.gather_rows(scores, idx.to(DEV))
    out.backward(w.to(DEV))
    acc = torch.zeros(B, T, dtype=torch.float32)
    acc.scatter_add_(1, torch.where(idx < 0, idx + T, idx), w.float())
    ref = acc.to(dtype).float().numpy()
    tol = 1e-6 if dtype == torch.float32 else 1e-2
    np.testing.assert_allclose(scores.grad.float().cpu().numpy(), ref, rtol=tol, atol=tol * np.abs(ref).max())


# ------------------------------------------------------------------------------------------------ CIF
@pytest.mark.parametrize("name", ["cif_plain", "cif_fixup", "cif_ragged"])
def test_cif_golden(ops, name):
    g = load(name)
    B, T = g["scores"].shape
    mask = lmask(g["lengths"], T)
    tl = dev(g["target_lengths"], torch.int32)
    scores = dev(g["scores"]).requires_grad_()
    alphas, q = ops.cif_quantity(scores, tl, attn_mask=dev(mask))
    np.testing.assert_allclose(alphas.cpu().numpy(), g["alphas"], rtol=1e-6, atol=1e-7)   # expf vs torch sigmoid
    np.testing.assert_allclose(q.item(), g["quantity_loss"], rtol=1e-5)
    q.backward()
    s_ref = torch.from_numpy(g["scores"]).requires_grad_()
    _, q_ref = orc.quantity_loss(s_ref, torch.from_numpy(mask), torch.from_numpy(g["target_lengths"]))
    q_ref.backward()
    np.testing.assert_allclose(scores.grad.cpu().numpy(), s_ref.grad.numpy(), rtol=1e-4, atol=1e-6)
    # resize on the reference's own alphas: fp32 row sums are order dependent -> rtol 1e-5
    resized = ops.cif_resize(dev(g["alphas"]), tl)
    np.testing.assert_allclose(resized.cpu().numpy(), g["resized"], rtol=1e-5, atol=1e-7)
    # fire scan on the reference's resized alphas: bit-exact positions
    for kw in (dict(attn_mask=dev(mask)), dict(lengths=dev(g["lengths"], torch.int32))):
        idx, nmask, counts, mx, _ = ops.cif_fire_scan(dev(g["resized"]), **kw)
        kf = int(mx.item())
        assert kf == g["nugget_indices"].shape[1]
        np.testing.assert_array_equal(idx[:, :kf].cpu().numpy(), g["nugget_indices"])
        np.testing.assert_array_equal(nmask[:, :kf].cpu().numpy(), g["nugget_mask"])
        assert (idx[:, kf:] == -1).all()
        np.testing.assert_array_equal(counts.cpu().numpy(), g["nugget_mask"].sum(1))
        block = ops.cif_block_mask(idx, counts, T=T, **kw)
        np.testing.assert_array_equal(block.cpu().numpy(), g["block_attn_mask"])


def test_scan_handmade(ops):
    g = load("cif_scan_handmade")
    T = g["alphas"].shape[1]
    mask = lmask(g["lengths"], T)
    idx, nmask, counts, mx, _ = ops.cif_fire_scan(dev(g["alphas"]), attn_mask=dev(mask))
    kf = int(mx.item())
    np.testing.assert_array_equal(idx[:, :kf].cpu().numpy(), g["nugget_indices"])
    block = ops.cif_block_mask(idx, counts, attn_mask=dev(mask))
    np.testing.assert_array_equal(block.cpu().numpy(), g["block_attn_mask"])


@pytest.mark.parametrize("B,T", [(64, 1500), (16, 3000), (70, 333), (1, 65)])
def test_scan_random_vs_oracle(ops, B, T):
    rng = np.random.default_rng(B + T)
    lengths = rng.integers(1, T + 1, size=B)
    lengths[0] = T
    mask = lmask(lengths, T)
    alphas = (rng.random((B, T)) * 0.3).astype(np.float32)
    alphas[1 % B, 5] = 3.7                         # carries through several frames
    o_mask, o_idx, _ = orc.fire_scan(alphas, mask, with_block_mask=False)
    scores = rng.standard_normal((B, T)).astype(np.float32)
    idx, nmask, counts, mx, sg = ops.cif_fire_scan(dev(alphas), lengths=dev(lengths, torch.int32), scores=dev(scores))
    kf = int(mx.item())
    assert kf == o_idx.shape[1]
    np.testing.assert_array_equal(idx[:, :kf].cpu().numpy(), o_idx)
    np.testing.assert_array_equal(nmask[:, :kf].cpu().numpy(), o_mask)
    ref_sg = scores[np.arange(B)[:, None], o_idx]
    np.testing.assert_array_equal(sg[:, :kf].cpu().numpy(), ref_sg)


def test_resize_random_vs_oracle(ops):
    rng = np.random.default_rng(7)
    B, T = 16, 700
    lengths = rng.integers(20, T + 1, size=B)
    mask = lmask(lengths, T)
    alphas = (1 / (1 + np.exp(-rng.standard_normal((B, T)) * 2 + 1))).astype(np.float32) * mask
    tl = np.maximum(2, (lengths * 0.075).astype(np.int64))
    tl[:4] = (lengths[:4] * 0.9).astype(np.int64)           # forces fix-up rounds
    ref = orc.resize(torch.from_numpy(alphas), torch.from_numpy(tl)).numpy()
    out = ops.cif_resize(dev(alphas), dev(tl, torch.int32)).cpu().numpy()
    np.testing.assert_allclose(out, ref, rtol=2e-5, atol=1e-7)


# ------------------------------------------------------------------------------------------------ scorer
def _scorer_case(B, T, d, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, T, d, generator=g)
    lim = 1 / d ** 0.5
    w1 = (torch.rand(d, d, generator=g) * 2 - 1) * lim
    b1 = (torch.rand(d, generator=g) * 2 - 1) * lim
    w2 = (torch.rand(1, d, generator=g) * 2 - 1) * lim
    b2 = (torch.rand(1, generator=g) * 2 - 1) * lim
    return x, w1, b1, w2, b2


@pytest.mark.parametrize("rows,cols", [(128, 192), (1024, 1024), (70, 33)])
def test_cast_transpose(ops, rows, cols):
    """fp32 -> bf16 copy + transposed copy (weight preparation of the scorer): bit-exact vs torch rounding."""
    from star_amd import _lib
    src = torch.randn(rows, cols, generator=torch.Generator().manual_seed(rows + cols)).to(DEV)
    dst = torch.empty(rows, cols, dtype=torch.bfloat16, device=DEV)
    dst_t = torch.empty(cols, rows, dtype=torch.bfloat16, device=DEV)
    _lib.check(_lib.lib().star_cast_transpose_bf16(_lib.ptr(src), rows, cols, _lib.ptr(dst), _lib.ptr(dst_t), None,
                                                   _lib.stream_ptr()), "star_cast_transpose_bf16")
    ref = src.to(torch.bfloat16)
    assert torch.equal(dst, ref) and torch.equal(dst_t, ref.t().contiguous())
    # with a row scale on the transposed copy only (the backward's (diag(w2) W1)^T): one rounding of the product
    sc = torch.randn(rows, generator=torch.Generator().manual_seed(rows)).to(DEV)
    _lib.check(_lib.lib().star_cast_transpose_bf16(_lib.ptr(src), rows, cols, _lib.ptr(dst), _lib.ptr(dst_t),
                                                   _lib.ptr(sc), _lib.stream_ptr()), "star_cast_transpose_bf16")
    assert torch.equal(dst, ref) and torch.equal(dst_t, (src * sc[:, None]).to(torch.bfloat16).t().contiguous())


@pytest.mark.parametrize("B,T,d", [(4, 250, 768), (2, 301, 1024), (1, 37, 128)])
def test_scorer_forward_backward(ops, B, T, d):
    x, w1, b1, w2, b2 = _scorer_case(B, T, d, B + T + d)
    # oracle on bf16-rounded x and W1 (what the MFMA path consumes), fp32 math
    xr, w1r = bf16_round(x).requires_grad_(), bf16_round(w1).requires_grad_()
    b1r, w2r, b2r = b1.clone().requires_grad_(), w2.clone().requires_grad_(), b2.clone().requires_grad_()
    s_ref = orc.scorer(xr, w1r, b1r, w2r, b2r)
    gs = torch.randn(B, T, generator=torch.Generator().manual_seed(1))
    # sparse cotangent as in the non-streaming path: only some frames receive a score gradient
    keep = torch.zeros(B, T, dtype=torch.bool)
    keep[:, ::4] = True
    gs_sparse = gs * keep
    (s_ref * gs_sparse).sum().backward()

    xd = x.to(torch.bfloat16).to(DEV).requires_grad_()
    params = [t.to(DEV).requires_grad_() for t in (w1, b1, w2, b2)]
    hint = ops.ActiveRows()
    s = ops.scorer(xd, *params, hint)
    # fp32 accumulation over d products of bf16 values: error ~ d * 2^-24 * |terms|; measured < 2e-4
    np.testing.assert_allclose(s.detach().cpu().numpy(), s_ref.detach().numpy(), rtol=2e-3, atol=2e-3)
    hint.set(torch.nonzero(keep.reshape(-1)).flatten().to(torch.int32).to(DEV))
    (s * gs_sparse.to(DEV)).sum().backward()
    # G = dH is rounded to bf16 before the dW1 / dX GEMMs: relative error ~ 2^-8 per term -> 2e-2 of the scale
    def close(a, b, what):
        a, b = a.float().cpu().numpy(), b.detach().numpy()
        scale = np.abs(b).max() + 1e-12
        assert np.abs(a - b).max() <= 2e-2 * scale, f"{what}: max err {np.abs(a - b).max()} vs scale {scale}"
    close(params[0].grad, w1r.grad, "dW1")
    close(params[1].grad, b1r.grad, "db1")
    close(params[2].grad, w2r.grad, "dw2")
    close(params[3].grad, b2r.grad, "db2")
    close(xd.grad, xr.grad, "dX")
    # rows without a score gradient must be exactly zero in dX
    assert (xd.grad.float().cpu()[~keep] == 0).all()


def test_scorer_dense_backward_matches_sparse_hint(ops):
    x, w1, b1, w2, b2 = _scorer_case(2, 130, 256, 3)
    gs = torch.randn(2, 130)
    outs = []
    for use_hint in (False, True):
        xd = x.to(torch.bfloat16).to(DEV).requires_grad_()
        params = [t.to(DEV).requires_grad_() for t in (w1, b1, w2, b2)]
        hint = ops.ActiveRows()
        s = ops.scorer(xd, *params, hint)
        if use_hint:
            hint.set(torch.arange(2 * 130, dtype=torch.int32, device=DEV))
        (s * gs.to(DEV)).sum().backward()
        outs.append([p.grad.clone() for p in params] + [xd.grad.clone()])
    for a, b in zip(*outs):
        assert torch.equal(a, b)


# ------------------------------------------------------------------------------------------------ attention
XATTN = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(G, "xattn_*.npz"))
               if "drop" not in os.path.basename(p))       # the given-mask dropout goldens pin the ORACLE (tests/test_gpu_round2.py)


def _pad_hd(t, hd):
    if t.shape[-1] == hd:
        return t
    return torch.nn.functional.pad(t, (0, hd - t.shape[-1]))


def _run_xattn(ops, q, k, v, sc, klen, ilk, gout, scale, use_key_len, hd_pad):
    """q,k,v,sc,gout fp32 CPU tensors -> kernel results as fp32 CPU tensors (inputs rounded to bf16)."""
    hd = q.shape[-1]
    qd, kd, vd = (_pad_hd(t, hd_pad).to(torch.bfloat16).to(DEV).requires_grad_() for t in (q, k, v))
    scd = sc.to(DEV).requires_grad_()
    bias = scd
    key_len = None
    kmask = torch.from_numpy(lmask(klen, k.shape[2]))
    if use_key_len:
        key_len = torch.tensor(np.asarray(klen), dtype=torch.int32, device=DEV)
    else:   # the drop-in's way: fold the additive finfo.min pad mask into the bias
        bias = scd + ((1.0 - kmask.float()) * torch.finfo(torch.float32).min).to(DEV)
    o, p = ops.xattn(qd, kd, vd, bias=bias, key_len=key_len, ilk_diag=2 if ilk else -1, scale=scale, need_probs=True)
    o.backward(_pad_hd(gout, hd_pad).to(torch.bfloat16).to(DEV))
    f = lambda t: t.float().cpu()[..., :hd]
    return f(o.detach()), p.cpu(), f(qd.grad), f(kd.grad), f(vd.grad), scd.grad.cpu()


@pytest.mark.parametrize("name", XATTN)
@pytest.mark.parametrize("use_key_len", [True, False])
def test_xattn_golden(ops, name, use_key_len):
    g = load(name)
    q, k, v, sc, gout = (torch.from_numpy(g[n]) for n in ("q", "k", "v", "scores", "gout"))
    hd = q.shape[-1]
    hd_pad = hd if hd in (64, 96, 128) else 64            # zero-padding the head dim changes nothing
    # oracle on the bf16-rounded operands
    qr, kr, vr = (bf16_round(t).requires_grad_() for t in (q, k, v))
    scr = sc.clone().requires_grad_()
    kmask = torch.from_numpy(lmask(g["key_lengths"], k.shape[2]))
    o_ref, p_ref = orc.xattn_core(qr, kr, vr, kmask, scr, bool(g["ilk"]))
    (o_ref * bf16_round(gout)).sum().backward()
    o, p, dq, dk, dv, dsc = _run_xattn(ops, q, k, v, sc, g["key_lengths"], bool(g["ilk"]), gout, 1 / hd ** 0.5,
                                       use_key_len, hd_pad)
    # bf16 P / dS operands and bf16 outputs: atol 2e-2, rtol 2e-2
    tol = dict(rtol=2e-2, atol=2e-2)
    np.testing.assert_allclose(o.numpy(), o_ref.detach().numpy(), **tol)
    np.testing.assert_allclose(p.numpy(), p_ref.detach().numpy(), rtol=1e-3, atol=1e-5)
    np.testing.assert_allclose(dq.numpy(), qr.grad.numpy(), **tol)
    np.testing.assert_allclose(dk.numpy(), kr.grad.numpy(), **tol)
    np.testing.assert_allclose(dv.numpy(), vr.grad.numpy(), **tol)
    np.testing.assert_allclose(dsc.numpy(), scr.grad.numpy(), **tol)
    # and the unrounded reference golden itself, at a bf16-input tolerance
    np.testing.assert_allclose(o.numpy(), g["o"], rtol=5e-2, atol=5e-2)
    # masked keys: exact zeros in P
    assert (p.numpy()[~np.broadcast_to(kmask.numpy()[:, None, None, :], p.shape)] == 0).all()


@pytest.mark.parametrize("B,h,L,K,hd,ilk", [(2, 8, 128, 375, 128, False), (2, 4, 24, 63, 96, False),
                                            (1, 8, 227, 229, 128, True), (2, 2, 130, 200, 64, True),
                                            (1, 2, 300, 750, 128, False),
                                            # L <= 128: barrier-free dK/dV kernel, one workgroup per (b, h) ...
                                            (25, 8, 100, 77, 64, False), (24, 8, 128, 200, 128, True),
                                            # ... B*h >= 192: the fused dQ/dK/dV kernel (rounds of 128 keys)
                                            (32, 8, 37, 300, 128, False), (26, 8, 128, 130, 64, True),
                                            (24, 8, 100, 700, 128, True), (48, 4, 16, 17, 64, False),
                                            # ... and with the key groups split over several workgroups
                                            (2, 2, 60, 200, 64, True), (1, 3, 128, 1000, 128, False)])
def test_xattn_random_vs_oracle(ops, B, h, L, K, hd, ilk):
    g = torch.Generator().manual_seed(B * 7 + L + K + hd)
    q = torch.randn(B, h, L, hd, generator=g)
    k = torch.randn(B, h, K, hd, generator=g)
    v = torch.randn(B, h, K, hd, generator=g)
    sc = torch.randn(B, K, generator=g)
    gout = torch.randn(B, h, L, hd, generator=g)
    klen = [K] + [max(3, K - 37 * (i + 1)) for i in range(B - 1)]
    qr, kr, vr = (bf16_round(t).requires_grad_() for t in (q, k, v))
    scr = sc.clone().requires_grad_()
    kmask = torch.from_numpy(lmask(klen, K))
    o_ref, p_ref = orc.xattn_core(qr, kr, vr, kmask, scr, ilk)
    (o_ref * bf16_round(gout)).sum().backward()
    o, p, dq, dk, dv, dsc = _run_xattn(ops, q, k, v, sc, klen, ilk, gout, 1 / hd ** 0.5, True, hd)

    def close(a, b, what, tol=2e-2):
        a, b = a.numpy(), b.detach().numpy()
        scale = np.abs(b).max()
        assert np.abs(a - b).max() <= tol * scale + 1e-6, f"{what}: max err {np.abs(a - b).max()} scale {scale}"
    close(o, o_ref, "O")
    close(p, p_ref, "P", 2e-3)
    close(dq, qr.grad, "dQ")
    close(dk, kr.grad, "dK")
    close(dv, vr.grad, "dV")
    close(dsc, scr.grad, "dscores")


@pytest.mark.parametrize("B,h,L,K,hd,nl", [(3, 4, 50, 90, 64, 3), (24, 8, 64, 140, 128, 2), (5, 1, 20, 33, 64, 2)])
def test_shared_scorer_logits_grad(ops, B, h, L, K, hd, nl):
    """share_scorer_logits: nl layers fed from one logits tensor; gradient = the per-layer path's (one launch sums
    the [B,h,K] partials of all layers instead of nl head reductions + autograd adds)."""
    from star_amd.nugget import score_biased_attn, share_scorer_logits
    g = torch.Generator().manual_seed(B + K)
    mk = lambda *s: torch.randn(*s, generator=g).to(torch.bfloat16).to(DEV)
    qs, ks, vs, gs = ([mk(B, h, n, hd) for _ in range(nl)] for n in (L, K, K, L))
    klen = [K] + [max(3, K - 11 * (i + 1)) for i in range(B - 1)]
    kmask = lmask(klen, K)
    add = dev(((1.0 - kmask[:, None, None, :].astype(np.float32)) * np.finfo(np.float32).min))
    sc0 = torch.randn(B, K, generator=g).to(DEV)
    res = []
    for shared in (False, True):
        sc = sc0.clone().requires_grad_()
        per_layer = share_scorer_logits(sc, nl, h, attention_mask=add) if shared else [sc] * nl
        outs = [score_biased_attn(qs[l], ks[l], vs[l], add, None, per_layer[l], limit_cross_attn=(l % 2 == 1))[0]
                for l in range(nl)]
        torch.autograd.backward(outs, gs)
        res.append((outs, sc.grad))
    for a, b in zip(res[0][0], res[1][0]):
        assert torch.equal(a, b)
    ref, got = res[0][1].cpu().numpy(), res[1][1].cpu().numpy()
    np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-6 * np.abs(ref).max())


def test_xattn_strided_views(ops):
    """K/V as [B,K,h,hd] -> permuted views (what split_heads produces) must give the same result as dense copies."""
    g = torch.Generator().manual_seed(5)
    B, h, L, K, hd = 2, 8, 40, 90, 128
    q = torch.randn(B, L, h, hd, generator=g).to(torch.bfloat16).to(DEV).permute(0, 2, 1, 3)
    kk = torch.randn(B, K, h, hd, generator=g).to(torch.bfloat16).to(DEV).permute(0, 2, 1, 3)
    vv = torch.randn(B, K, h, hd, generator=g).to(torch.bfloat16).to(DEV).permute(0, 2, 1, 3)
    sc = torch.randn(B, K, generator=g).to(DEV)
    o1, _ = ops.xattn(q, kk, vv, bias=sc)
    o2, _ = ops.xattn(q.contiguous(), kk.contiguous(), vv.contiguous(), bias=sc)
    assert torch.equal(o1, o2)


# ------------------------------------------------------------------------------------------------ drop-in mirror
def test_dropin_signatures(ops):
    """The reference-named methods (NuggetOpsMixin) reproduce the golden call sites."""
    from star_amd.nugget import NuggetOpsMixin, score_biased_attn

    class Host(NuggetOpsMixin):
        nugget_compress_rate = 4
        nugget_model = None

    host = Host()
    g = load("select_c1")
    T = g["scores"].shape[1]
    mask = dev(lmask(g["lengths"], T))
    nmask, idx, sg = host.apply_nugget_compression(torch.zeros(4, T, 2, device=DEV), mask, nugget_scores=dev(g["scores"]))
    np.testing.assert_array_equal(idx.cpu().numpy(), g["indices"])
    np.testing.assert_array_equal(nmask.cpu().numpy(), g["nugget_mask"])
    table, pos = host.add_nugget_type_embedding(torch.zeros(4, T, 2, device=DEV), idx, nmask)
    np.testing.assert_array_equal(table.cpu().numpy(), g["type_table"])
    np.testing.assert_array_equal(pos.cpu().numpy(), g["positions"])

    c = load("cif_fixup")
    T = c["scores"].shape[1]
    mask = dev(lmask(c["lengths"], T))
    resized = host.resize(dev(c["alphas"]), dev(c["target_lengths"]))
    np.testing.assert_allclose(resized.cpu().numpy(), c["resized"], rtol=1e-5, atol=1e-7)
    nm, ni, blk = host.apply_threshold_yield_nugget(dev(c["resized"]), mask)
    np.testing.assert_array_equal(ni.cpu().numpy(), c["nugget_indices"])
    np.testing.assert_array_equal(nm.cpu().numpy(), c["nugget_mask"])
    np.testing.assert_array_equal(blk.cpu().numpy(), c["block_attn_mask"])
    vi, mv, nn_ = host.get_valid_mask_indices(dev(c["nugget_mask"]))
    assert mv == int(c["nugget_mask"].sum(1).max())

    x = load("xattn_hd96")
    q, k, v = (dev(x[n], torch.bfloat16) for n in ("q", "k", "v"))
    kmask = lmask(x["key_lengths"], k.shape[2])
    add = dev(((1.0 - kmask[:, None, None, :].astype(np.float32)) * np.finfo(np.float32).min))
    o, p = score_biased_attn(q, k, v, add, None, dev(x["scores"]), limit_cross_attn=bool(x["ilk"]), output_attentions=True)
    np.testing.assert_allclose(o.float().cpu().numpy(), x["o"], rtol=5e-2, atol=5e-2)
    np.testing.assert_allclose(p.cpu().numpy(), x["p"], rtol=5e-2, atol=5e-3)


def test_reference_style_composition_fp32(ops):
    """The drop-in classes composed the way W2VASRModel.lm_nugget_training does (w2v_asr.py:553-578), in fp32 like
    the reference runs: scorer -> apply_nugget_compression(nugget_scores=None path too) -> type table -> type
    embedding -> row gather -> cross-attention module with scorer_logits -> loss -> backward into every parameter.
    Compared with the same chain restated with the oracle (bf16 arithmetic inside the kernels: 3e-2 of the scale)."""
    from star_amd.nugget import (NuggetCrossAttention, NuggetOpsMixin, NuggetScorer, gather_nuggets)
    torch.manual_seed(21)
    B, T, d, h, L, rate = 3, 48, 256, 4, 10, 4
    lengths = [48, 37, 9]

    class Host(NuggetOpsMixin, torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.nugget_compress_rate = rate
            self.nugget_model = NuggetScorer(d)
            self.type_embed = torch.nn.Embedding(2, d)
            self.crossattention = NuggetCrossAttention(d, h)

    m = Host().to(DEV)
    with torch.no_grad():
        for prm in m.crossattention.parameters():
            prm.copy_(torch.randn_like(prm) * (0.05 if prm.dim() == 2 else 0.02))
    x = torch.randn(B, T, d, device=DEV)
    x = ((x - x.mean(-1, keepdim=True)) / x.std(-1, keepdim=True)).requires_grad_()
    hid = torch.randn(B, L, d, device=DEV)
    mask = dev(lmask(lengths, T))
    gout = torch.randn(B, L, d, device=DEV)

    nmask, idx, sg = m.apply_nugget_compression(x, mask)                       # scores from self.nugget_model
    scores = m.nugget_model(x).squeeze(-1)
    nmask2, idx2, sg2 = m.apply_nugget_compression(x, mask, nugget_scores=scores)
    assert torch.equal(idx, idx2) and torch.equal(nmask, nmask2)
    table, pos = m.add_nugget_type_embedding(x, idx2, nmask2)
    emb = m.add_type_embedding(x, table, m.type_embed)
    sem = torch.tanh(emb)                                                      # stand-in for the semantic encoder
    mem = gather_nuggets(sem, idx2)
    add = (1.0 - nmask2[:, None, None, :].float()) * torch.finfo(torch.float32).min
    out = m.crossattention(hid, encoder_hidden_states=mem, encoder_attention_mask=add, scorer_logits=sg2)[0]
    assert out.dtype == torch.float32
    (out * gout).sum().backward()

    # oracle restatement (fp32, CPU) on the same parameters
    P = {n: t.detach().cpu().clone().requires_grad_() for n, t in m.named_parameters()}
    xr = x.detach().cpu().clone().requires_grad_()
    s_ref = orc.scorer(xr, P["nugget_model.0.weight"], P["nugget_model.0.bias"], P["nugget_model.2.weight"],
                       P["nugget_model.2.bias"])
    s_ref.retain_grad()
    # index parity on scores produced by the same scorer implementation: select the GPU's scores
    o_mask, o_idx, _ = orc.topk_select(scores.detach().cpu().numpy(), lmask(lengths, T), rate=float(rate))
    np.testing.assert_array_equal(idx2.cpu().numpy(), o_idx)
    np.testing.assert_array_equal(nmask2.cpu().numpy(), o_mask)
    o_table, o_pos = orc.type_table(T, o_idx, o_mask)
    np.testing.assert_array_equal(table.cpu().numpy(), o_table)
    np.testing.assert_array_equal(pos.cpu().numpy(), o_pos)
    it = torch.from_numpy(o_idx)
    sg_ref = s_ref.gather(1, it)
    sem_r = torch.tanh(orc.type_embed_add(xr, torch.from_numpy(o_table), P["type_embed.weight"]))
    mem_r = orc.gather_rows(sem_r, it)
    split = lambda t: t.view(t.shape[0], t.shape[1], h, d // h).permute(0, 2, 1, 3)
    q = split(hid.cpu() @ P["crossattention.q_attn.weight"] + P["crossattention.q_attn.bias"])
    kv = mem_r @ P["crossattention.c_attn.weight"] + P["crossattention.c_attn.bias"]
    o_ref, _ = orc.xattn_core(q, split(kv[..., :d]), split(kv[..., d:]), torch.from_numpy(o_mask), sg_ref, False)
    y = o_ref.permute(0, 2, 1, 3).reshape(B, L, d) @ P["crossattention.c_proj.weight"] + P["crossattention.c_proj.bias"]
    (y * gout.cpu()).sum().backward()

    def close(a, b, what, tol=3e-2):
        a, b = a.detach().float().cpu().numpy(), b.detach().numpy()
        scale = np.abs(b).max()
        assert np.abs(a - b).max() <= tol * scale + 1e-6, f"{what}: max err {np.abs(a - b).max()} scale {scale}"
    close(scores, s_ref, "scores", 5e-3)
    close(out, y, "output")
    close(x.grad, xr.grad, "d X")
    for n, t in m.named_parameters():
        assert t.grad is not None, n
        if n.startswith("nugget_model.0"):
            # the ReLU gate of a hidden unit flips under bf16 where its pre-activation is ~0: the whole dW1 row /
            # db1 entry of that unit then moves by one frame's term.  Allow 2% of such entries, as in the e2e test.
            a, b = t.grad.detach().cpu().numpy(), P[n].grad.numpy()
            bad = np.abs(a - b) > 3e-2 * np.abs(b).max() + 1e-6
            assert bad.mean() <= 0.02, f"d {n}: {bad.mean():.3%} of the entries off"
        elif n == "nugget_model.2.bias":
            # db2 = sum of dscores: analytically 0 here (softmax is shift invariant), so bound it by the terms
            assert abs(float(t.grad) - float(P[n].grad)) <= 5e-3 * float(s_ref.grad.abs().sum())
        else:
            close(t.grad, P[n].grad, "d " + n)


def test_reference_style_composition_simul(ops):
    """The simultaneous path composed as W2VSimulModel.lm_nugget_training does (w2v_simul.py:407-465), with the
    torch lines a maintainer keeps (sigmoid, quantity loss, indexing with -1) next to the drop-in methods (resize,
    apply_threshold_yield_nugget, add_nugget_type_embedding, gather, ILK cross-attention)."""
    from star_amd.nugget import NuggetCrossAttention, NuggetOpsMixin, NuggetScorer, gather_nuggets
    torch.manual_seed(23)
    B, T, d, h, L = 3, 60, 256, 4, 9
    lengths, tlen = [60, 44, 13], [7, 5, 2]

    class Host(NuggetOpsMixin, torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.nugget_model = NuggetScorer(d)
            self.crossattention = NuggetCrossAttention(d, h)

    m = Host().to(DEV)
    with torch.no_grad():
        for prm in m.crossattention.parameters():
            prm.copy_(torch.randn_like(prm) * (0.05 if prm.dim() == 2 else 0.02))
    x = torch.randn(B, T, d, device=DEV)
    x = (x - x.mean(-1, keepdim=True)) / x.std(-1, keepdim=True)
    sem = torch.randn(B, T, d, device=DEV, requires_grad=True)
    hid = torch.randn(B, L, d, device=DEV)
    gout = torch.randn(B, L, d, device=DEV)
    mask = dev(lmask(lengths, T))
    tl = torch.tensor(tlen, device=DEV)

    scores = m.nugget_model(x.detach()).squeeze(-1)                                       # :410-411
    alphas = torch.sigmoid(scores) * mask                                                 # :419
    qloss = ((alphas.sum(1) / 1.0 - tl) ** 2).mean()                                      # :420-424
    resized = m.resize(alphas, tl)                                                        # :425
    nmask, nidx, _ = m.apply_threshold_yield_nugget(resized, mask)                        # :426
    nscores = scores[torch.arange(B, device=DEV).unsqueeze(1), nidx]                      # :427 (-1 = last frame)
    table, pos = m.add_nugget_type_embedding(x, nidx, nmask)                              # :431
    mem = gather_nuggets(sem, nidx)                                                       # :446-447
    add = (1.0 - nmask[:, None, None, :].float()) * torch.finfo(torch.float32).min
    out = m.crossattention(hid, encoder_hidden_states=mem, encoder_attention_mask=add, scorer_logits=nscores,
                           limit_cross_attn=True)[0]
    ((out * gout).sum() + 0.01 * qloss).backward()                                        # :461

    # oracle chain on the GPU's scores (index parity is asserted on scores of the same scorer implementation)
    P = {n: t.detach().cpu().clone().requires_grad_() for n, t in m.named_parameters()}
    s_leaf = scores.detach().cpu().clone().requires_grad_()
    mk = torch.from_numpy(lmask(lengths, T))
    a_ref = torch.sigmoid(s_leaf) * mk
    q_ref = ((a_ref.sum(1) - torch.tensor(tlen)) ** 2).mean()
    r_ref = orc.resize(a_ref.detach(), torch.tensor(tlen))
    np.testing.assert_allclose(resized.detach().cpu().numpy(), r_ref.numpy(), rtol=2e-5, atol=1e-7)
    f_mask, f_idx, _ = orc.fire_scan(resized.detach().cpu().numpy(), lmask(lengths, T), with_block_mask=False)
    kf = nidx.shape[1]
    np.testing.assert_array_equal(nidx.cpu().numpy(), f_idx[:, :kf])
    np.testing.assert_array_equal(nmask.cpu().numpy(), f_mask[:, :kf])
    it = torch.from_numpy(f_idx[:, :kf])
    ns_ref = s_leaf[torch.arange(B).unsqueeze(1), it]
    sem_r = sem.detach().cpu().clone().requires_grad_()
    mem_r = orc.gather_rows(sem_r, it)
    split = lambda t: t.view(t.shape[0], t.shape[1], h, d // h).permute(0, 2, 1, 3)
    q = split(hid.cpu() @ P["crossattention.q_attn.weight"] + P["crossattention.q_attn.bias"])
    kv = mem_r @ P["crossattention.c_attn.weight"] + P["crossattention.c_attn.bias"]
    o_ref, _ = orc.xattn_core(q, split(kv[..., :d]), split(kv[..., d:]), torch.from_numpy(f_mask[:, :kf]), ns_ref, True)
    y = o_ref.permute(0, 2, 1, 3).reshape(B, L, d) @ P["crossattention.c_proj.weight"] + P["crossattention.c_proj.bias"]
    ((y * gout.cpu()).sum() + 0.01 * q_ref).backward()

    def close(a, b, what, tol=3e-2):
        a, b = a.detach().float().cpu().numpy(), b.detach().numpy()
        scale = np.abs(b).max()
        assert np.abs(a - b).max() <= tol * scale + 1e-6, f"{what}: max err {np.abs(a - b).max()} scale {scale}"
    close(out, y, "output")
    close(sem.grad, sem_r.grad, "d semantic states")
    for n in ("q_attn", "c_attn", "c_proj"):
        close(getattr(m.crossattention, n).weight.grad, P[f"crossattention.{n}.weight"].grad, "d " + n)
    # the scorer gradients against torch's own backward through the scorer from the oracle's dscores
    ds = s_leaf.grad.to(DEV)
    ref_params = [t.detach().clone().requires_grad_() for t in m.nugget_model.parameters()]
    w1, b1, w2, b2 = ref_params
    s2 = (torch.relu(x @ w1.t() + b1) @ w2.t() + b2).squeeze(-1)
    s2.backward(ds)
    for (n, t), r in zip(m.nugget_model.named_parameters(), ref_params):
        a, b = t.grad.detach().cpu().numpy(), r.grad.cpu().numpy()
        bad = np.abs(a - b) > 3e-2 * np.abs(b).max() + 1e-6
        assert bad.mean() <= 0.02, f"d nugget_model.{n}: {bad.mean():.3%} of the entries off"


@pytest.mark.parametrize("ilk", [False, True])
def test_cross_attention_module_vs_oracle(ops, ilk):
    """Row a9: the whole cross-attention branch of MyGPT2Attention.forward (my_transformer.py:490-533) -- q_attn,
    c_attn, split heads, score-biased core, merge heads, c_proj -- against the same arithmetic in fp32 with the
    oracle's core, forward and parameter / input gradients."""
    from star_amd.nugget import NuggetCrossAttention
    torch.manual_seed(11)
    d, h, B, L, K = 256, 4, 3, 20, 37
    m = NuggetCrossAttention(d, h).to(DEV, torch.bfloat16)
    with torch.no_grad():
        for prm in m.parameters():
            prm.copy_(torch.randn_like(prm) * (0.05 if prm.dim() == 2 else 0.02))
    x = torch.randn(B, L, d, device=DEV).to(torch.bfloat16).requires_grad_()
    enc = torch.randn(B, K, d, device=DEV).to(torch.bfloat16).requires_grad_()
    sc = torch.randn(B, K, device=DEV).requires_grad_()
    klen = [K, K - 9, 5]
    kmask = lmask(klen, K)
    add = dev(((1.0 - kmask[:, None, None, :].astype(np.float32)) * np.finfo(np.float32).min))
    gout = torch.randn(B, L, d, device=DEV).to(torch.bfloat16)
    out = m(x, encoder_hidden_states=enc, encoder_attention_mask=add, scorer_logits=sc, limit_cross_attn=ilk)[0]
    out.backward(gout)

    # fp32 restatement on the same (bf16-valued) parameters and inputs
    P = {n: t.detach().float().cpu().requires_grad_() for n, t in m.named_parameters()}
    xr, er, sr = (t.detach().float().cpu().requires_grad_() for t in (x, enc, sc))
    split = lambda t: t.view(t.shape[0], t.shape[1], h, d // h).permute(0, 2, 1, 3)
    q = split(xr @ P["q_attn.weight"] + P["q_attn.bias"])
    kv = er @ P["c_attn.weight"] + P["c_attn.bias"]
    k, v = split(kv[..., :d]), split(kv[..., d:])
    o_ref, _ = orc.xattn_core(q, k, v, torch.from_numpy(kmask), sr, ilk)
    y = o_ref.permute(0, 2, 1, 3).reshape(B, L, d) @ P["c_proj.weight"] + P["c_proj.bias"]
    (y * gout.float().cpu()).sum().backward()

    def close(a, b, what, tol=3e-2):
        a, b = a.detach().float().cpu().numpy(), b.detach().numpy()
        scale = np.abs(b).max()
        assert np.abs(a - b).max() <= tol * scale + 1e-6, f"{what}: max err {np.abs(a - b).max()} scale {scale}"
    close(out, y, "output")
    close(x.grad, xr.grad, "d hidden")
    close(enc.grad, er.grad, "d encoder memory")
    close(sc.grad, sr.grad, "d scorer_logits")
    for n, t in m.named_parameters():
        close(t.grad, P[n].grad, "d " + n)


def test_cross_attention_kv_cache(ops):
    """Generation keeps the projected encoder K/V: the reference recomputes c_attn(memory) for every
    token (my_transformer.py:724-733); the cached path must give the same output and project the memory once."""
    from star_amd.nugget import NuggetCrossAttention
    torch.manual_seed(3)
    d, h, B, K = 512, 8, 3, 40
    m = NuggetCrossAttention(d, h).to(DEV, torch.bfloat16)
    calls = []
    m.c_attn.register_forward_hook(lambda *a: calls.append(1))
    enc = torch.randn(B, K, d, device=DEV, dtype=torch.bfloat16)
    sc = torch.randn(B, K, device=DEV)
    steps = [torch.randn(B, 1, d, device=DEV, dtype=torch.bfloat16) for _ in range(3)]

    m.train()
    ref = [m(x, encoder_hidden_states=enc, scorer_logits=sc)[0].detach() for x in steps]
    assert len(calls) == 3 and m._kv_cache is None          # training never caches

    m.eval()
    calls.clear()
    with torch.no_grad():
        got = [m(x, encoder_hidden_states=enc, scorer_logits=sc)[0] for x in steps]
        assert len(calls) == 1
        for a, b in zip(got, ref):
            assert torch.equal(a, b)
        enc.mul_(0.5)                                         # in-place change of the memory invalidates the entry
        o1 = m(steps[0], encoder_hidden_states=enc, scorer_logits=sc)[0]
        assert len(calls) == 2 and not torch.equal(o1, ref[0])
        enc2 = enc.clone()                                    # another tensor with equal contents: recomputed, equal
        o2 = m(steps[0], encoder_hidden_states=enc2, scorer_logits=sc)[0]
        assert len(calls) == 3 and torch.equal(o1, o2)
        m.c_attn.weight.mul_(1.5)                             # weights changed (e.g. a checkpoint load)
        m(steps[0], encoder_hidden_states=enc2, scorer_logits=sc)
        assert len(calls) == 4
        m.clear_kv_cache()
        m(steps[0], encoder_hidden_states=enc2, scorer_logits=sc)
        assert len(calls) == 5
    m(steps[0], encoder_hidden_states=enc2, scorer_logits=sc)  # eval but grad enabled: no cache
    assert len(calls) == 6 and m._kv_cache is None


# ------------------------------------------------------------------------------------------------ whole step
def test_nugget_step_asr_vs_oracle(ops):
    from star_amd.nugget import NuggetScorer
    from star_amd.step import make_inputs, nugget_step_asr
    torch.manual_seed(0)
    lengths, T, d, h, L, nl, rate = [250, 230, 200, 170], 250, 768, 8, 24, 2, 4
    scorer = NuggetScorer(d).to(DEV)
    inp = make_inputs(lengths, T, d, h, L, nl, DEV)
    out = nugget_step_asr(scorer, inp, rate, with_loss=True)
    # index parity on the scores produced by the SAME scorer implementation
    scores = out["scores"].detach().cpu().numpy()
    mask = lmask(lengths, T)
    o_mask, o_idx, o_sg = orc.topk_select(scores, mask, rate)
    np.testing.assert_array_equal(out["indices"].cpu().numpy(), o_idx)
    np.testing.assert_array_equal(out["nugget_mask"].cpu().numpy(), o_mask)
    # the whole step in fp32 on the CPU, from the same bf16 inputs
    cpu = lambda t: t.detach().float().cpu()
    w1, b1, w2, b2 = (cpu(p).requires_grad_() for p in (scorer[0].weight, scorer[0].bias, scorer[2].weight, scorer[2].bias))
    w1r = bf16_round(w1.detach()).requires_grad_()
    xr, sr = cpu(inp.x).requires_grad_(), cpu(inp.s).requires_grad_()
    qs = [cpu(q).requires_grad_() for q in inp.qs]
    # the oracle step runs on the kernel's selection (bit-exact above on the kernel's own scores), so that a near-tie
    # resolved differently by the fp32 scorer cannot skip the comparison of everything downstream
    ref = orc.nugget_step_asr(xr, sr, lengths, rate, w1r, b1, w2, b2, qs, [(cpu(a), cpu(b)) for a, b in inp.kv_scales],
                              [cpu(g) for g in inp.gs], select=(o_mask, o_idx))
    # the step scores a ragged batch through star_scorer_fwd_ragged: pad frames (masked by every consumer) are 0
    np.testing.assert_allclose(scores[mask], ref["scores"].detach().numpy()[mask], rtol=2e-3, atol=2e-3)
    assert (scores[~mask] == 0).all()
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
    close(inp.s.grad, sr.grad, "dS")
    close(inp.qs[0].grad, qs[0].grad, "dQ0")
    close(inp.x.grad, xr.grad, "dX")


def test_nugget_step_simul_runs_and_matches_oracle_pieces(ops):
    from star_amd.nugget import NuggetScorer
    from star_amd.step import make_inputs, nugget_step_simul
    torch.manual_seed(1)
    lengths, T, d, h, nl = [300, 280, 150, 60], 300, 256, 2, 2
    tl = [int(round(0.075 * n)) + 2 for n in lengths]
    L = max(tl)
    scorer = NuggetScorer(d).to(DEV)
    inp = make_inputs(lengths, T, d, h, L, nl, DEV, target_lengths=tl)
    for ilk in (False, True):
        scorer.zero_grad()
        out = nugget_step_simul(scorer, inp, use_ilk=ilk, with_loss=True)
        mask = lmask(lengths, T)
        # fire positions: bit-exact against the oracle scan run on the kernel's own resized alphas
        o_mask, o_idx, _ = orc.fire_scan(out["resized"].cpu().numpy(), mask, with_block_mask=False)
        kf = int(out["max_count"].item())
        assert kf == o_idx.shape[1] and kf <= out["indices"].shape[1]
        np.testing.assert_array_equal(out["indices"][:, :kf].cpu().numpy(), o_idx)
        ref_resized = orc.resize(out["alphas"].cpu(), torch.tensor(tl))
        np.testing.assert_allclose(out["resized"].cpu().numpy(), ref_resized.numpy(), rtol=2e-5, atol=1e-7)
        assert torch.isfinite(out["loss"]).item()
        assert scorer[0].weight.grad is not None and torch.isfinite(scorer[0].weight.grad).all()
        assert inp.x.grad is None                      # the simul path detaches X (w2v_simul.py:410)


# ------------------------------------------------------------------------------------------------ e2e call sites
@pytest.mark.parametrize("name", ["e2e_asr", "e2e_simul", "e2e_simul_ilk"])
def test_e2e_call_sites_gpu(ops, name):
    """Whole-model run of the reference (fixtures from tests/golden/gen_golden.py): the
    HIP path, chained over the recorded call-site tensors exactly like the reference chains its own ops, must give
    the reference's selection (bit-exact, on the reference's scores), every cross-attention output and gradient,
    the total scorer-logit gradient and the nugget_model parameter gradients (bf16-MFMA bars of section 2)."""
    from star_amd.nugget import NuggetScorer
    from tests.test_oracle_golden import e2e_w1
    g = load(name)
    simul = "quantity_loss" in g
    B, T, d = g["x"].shape
    w1 = e2e_w1(g)
    scorer = NuggetScorer(d).to(DEV)
    with torch.no_grad():
        scorer[0].weight.copy_(dev(w1))
        scorer[0].bias.copy_(dev(g["b1"]))
        scorer[2].weight.copy_(dev(g["w2"]))
        scorer[2].bias.copy_(dev(g["b2"]))
    x = dev(g["x"], torch.bfloat16)
    scores = scorer(x).squeeze(-1).float()
    np.testing.assert_allclose(scores.detach().cpu().numpy(), g["scores"], atol=2e-2, rtol=2e-2)
    # selection on the reference's own scores: integer-exact
    lengths = dev(g["lengths"], torch.int32)
    tl = dev(g["target_lengths"], torch.int32)
    ref_scores = dev(g["scores"])
    if simul:
        alphas, _ = ops.cif_quantity(ref_scores, tl, lengths=lengths)
        resized = ops.cif_resize(alphas, tl)
        idx, nmask, counts, _, _ = ops.cif_fire_scan(resized, lengths=lengths, cap=g["nugget_position"].shape[1])
        key_len = counts
    else:
        nmask, idx, _, table, pos = ops.topk_select(ref_scores, lengths=lengths, rate=float(g["rate"]))
        key_len = nmask.sum(1).to(torch.int32)
    pos = torch.where(nmask.bool(), idx, torch.full_like(idx, -1))
    np.testing.assert_array_equal(pos.cpu().numpy(), g["nugget_position"])
    # differentiable chain on the HIP scores: gather -> nl x core -> sum <O, dO> (+ 0.01 quantity loss)
    sg = ops.gather_rows(scores, pos if simul else idx)
    total = 0.0
    qkv = []
    for l in range(int(g["n_layers"])):
        q, k, v = (dev(g[f"{n}{l}"], torch.bfloat16).requires_grad_() for n in "qkv")
        o, _ = ops.xattn(q, k, v, bias=sg, key_len=key_len, ilk_diag=2 if int(g["use_ilk"]) else -1)
        ref_o = g[f"o{l}"]
        np.testing.assert_allclose(o.detach().float().cpu().numpy(), ref_o, atol=2e-2 * np.abs(ref_o).max(), rtol=2e-2)
        total = total + (o.float() * dev(g[f"do{l}"])).sum()
        qkv.append((q, k, v))
    if simul:
        _, ql = ops.cif_quantity(scores, tl, lengths=lengths)
        np.testing.assert_allclose(ql.item(), float(g["quantity_loss"]), rtol=2e-2)
        total = total + 0.01 * ql
    scores.retain_grad()
    total.backward()

    def close(got, ref, what, gate_flips=0.0):
        """|got - ref| <= 2e-2 * (max|ref| + |ref|).  gate_flips: admitted fraction of outliers for dW1 / db1, whose
        terms ds*w2*[h > 0] are gated by the sign of h: a pre-activation within bf16 rounding of 0 flips its gate when
        X and W1 are rounded to bf16, which moves that element by one whole term (measured: < 1 % of the elements)."""
        ref = np.asarray(ref)
        got = got.detach().float().cpu().numpy().reshape(ref.shape)
        bad = np.abs(got - ref) > 2e-2 * np.abs(ref).max() + 2e-2 * np.abs(ref)
        assert bad.mean() <= gate_flips, f"{what}: {bad.sum()} of {bad.size} elements off"

    for l, (q, k, v) in enumerate(qkv):
        close(q.grad, g[f"dq{l}"], f"dq{l}")
        close(k.grad, g[f"dk{l}"], f"dk{l}")
        close(v.grad, g[f"dv{l}"], f"dv{l}")
    close(scores.grad, g["dscores"], "dscores")
    close(scorer[0].weight.grad[:64], g["dw1_rows64"], "dW1", gate_flips=0.02)
    close(scorer[0].bias.grad, g["db1"], "db1", gate_flips=0.02)
    close(scorer[2].weight.grad, g["dw2"], "dw2")
    ref_db2 = np.asarray(g["db2"])
    # db2 = sum of dscores, analytically 0 in the ASR path (softmax is shift invariant): the bf16 rounding errors of
    # the individual terms do not cancel, so the bar is relative to sum |dscores| (measured 1.3e-3 of it)
    np.testing.assert_allclose(scorer[2].bias.grad.cpu().numpy(), ref_db2, rtol=2e-2,
                               atol=5e-3 * np.abs(g["dscores"]).sum())


# ------------------------------------------------------------------------------------------------ 8f-2: CIF aggregation
@pytest.mark.parametrize("name", ["cifagg_201", "cifagg_202"])
def test_cif_aggregate_golden_gpu(ops, name):
    """Reference call site lm_cif_training (w2v_simul.py:558-606): positions / mask integer-exact; fp32 states: output
    within 1 ulp (same accumulation order), gradients rtol 1e-4 (summation order differs)."""
    from star_amd.nugget import NuggetOpsMixin
    g = load(name)
    alphas = dev(g["alphas"]).requires_grad_()
    s = dev(g["s"]).requires_grad_()
    cif, mask, pos = NuggetOpsMixin.cif_weighted_aggregate(alphas, s)
    np.testing.assert_array_equal(pos.cpu().numpy(), g["positions"])
    np.testing.assert_array_equal(mask.cpu().numpy(), g["cif_mask"])
    np.testing.assert_allclose(cif.detach().cpu().numpy(), g["cif"], rtol=1e-6, atol=2e-7)     # measured: <= 1 ulp
    cif.backward(dev(g["dcif"]))
    np.testing.assert_allclose(s.grad.cpu().numpy(), g["ds"], rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(alphas.grad.cpu().numpy(), g["dalphas"], rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize("B,T,d,dtype", [(5, 300, 768, torch.bfloat16), (70, 129, 64, torch.float32),
                                         (2, 1500, 1024, torch.bfloat16), (3, 65, 100, torch.float32)])
def test_cif_aggregate_random_vs_oracle(ops, B, T, d, dtype):
    rng = np.random.default_rng(B + T + d)
    lengths = rng.integers(1, T + 1, size=B)
    lengths[0] = T
    mask = lmask(lengths, T)
    alphas_np = (rng.random((B, T)).astype(np.float32) * 0.4) * mask        # a fire every ~5 frames
    alphas_np[B - 1] *= 0.001                                                # a row that never fires
    s_np = rng.standard_normal((B, T, d)).astype(np.float32)
    s_ref = torch.from_numpy(s_np).to(dtype).float().requires_grad_()        # oracle on the rounded states
    a_ref = torch.from_numpy(alphas_np).clone().requires_grad_()
    o_cif, o_mask, o_pos = orc.cif_aggregate(a_ref, s_ref)
    a = dev(alphas_np).requires_grad_()
    s = dev(s_np, dtype).requires_grad_()
    cif, cmask, idx, counts, mx = ops.cif_aggregate(a, s, cap=T)
    kf = o_pos.shape[1]
    assert int(mx.item()) == kf
    np.testing.assert_array_equal(idx[:, :kf].cpu().numpy(), o_pos.numpy())
    assert (idx[:, kf:] == -1).all() and (cif[:, kf:] == 0).all()
    np.testing.assert_array_equal(cmask[:, :kf].cpu().numpy(), o_mask.numpy().astype(np.uint8))
    tol = dict(rtol=1e-2, atol=1e-2) if dtype == torch.bfloat16 else dict(rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(cif[:, :kf].detach().float().cpu().numpy(), o_cif.detach().numpy(), **tol)
    gout = torch.from_numpy(rng.standard_normal((B, kf, d)).astype(np.float32)).to(dtype)
    (o_cif * gout.float()).sum().backward()
    gfull = torch.zeros(B, T, d, dtype=dtype)
    gfull[:, :kf] = gout
    cif.backward(gfull.to(DEV))
    gtol = dict(rtol=2e-2, atol=2e-2 * float(s_ref.grad.abs().max())) if dtype == torch.bfloat16 else dict(rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(s.grad.float().cpu().numpy(), s_ref.grad.numpy(), **gtol)
    atol_a = (2e-2 if dtype == torch.bfloat16 else 1e-4) * float(a_ref.grad.abs().max())
    np.testing.assert_allclose(a.grad.cpu().numpy(), a_ref.grad.numpy(), rtol=2e-2 if dtype == torch.bfloat16 else 1e-4,
                               atol=atol_a)


# ------------------------------------------------------------------------------------------------ 8f-3: type embedding
@pytest.mark.parametrize("name", ["typeembed_a", "typeembed_b"])
def test_type_embed_golden_gpu(ops, name):
    """x + type_embed(table) through the C ABI: forward bit-exact (fp32), d x = d out, d emb rtol 1e-5."""
    from star_amd.nugget import NuggetOpsMixin
    g = load(name)
    x = dev(g["x"]).requires_grad_()
    emb = torch.nn.Embedding(2, x.shape[-1]).to(DEV)
    with torch.no_grad():
        emb.weight.copy_(dev(g["emb"]))
    out = NuggetOpsMixin.add_type_embedding(x, dev(g["table"]), emb)
    np.testing.assert_array_equal(out.detach().cpu().numpy(), g["out"])
    (out * dev(g["gout"])).sum().backward()
    np.testing.assert_array_equal(x.grad.cpu().numpy(), g["dx"])
    np.testing.assert_allclose(emb.weight.grad.cpu().numpy(), g["demb"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("B,T,d,dtype", [(32, 1500, 1024, torch.bfloat16), (3, 77, 200, torch.float32),
                                         (5, 130, 520, torch.bfloat16)])
def test_type_embed_random_vs_oracle(ops, B, T, d, dtype):
    g = torch.Generator().manual_seed(B + T + d)
    x = torch.randn(B, T, d, generator=g).to(dtype)
    w = (torch.randn(2, d, generator=g) * 0.1).to(dtype)
    table = (torch.rand(B, T, generator=g) < 0.25).long()
    gout = torch.randn(B, T, d, generator=g).to(dtype)
    xr, wr = x.float().clone().requires_grad_(), w.float().clone().requires_grad_()
    ref = orc.type_embed_add(xr, table, wr)
    (ref * gout.float()).sum().backward()
    xd, wd = x.detach().to(DEV).requires_grad_(), w.detach().to(DEV).requires_grad_()
    out = ops.type_embed_add(xd, table.to(DEV), wd)
    # one rounding of the fp32 sum: bit-exact against the oracle's fp32 sum cast once
    assert torch.equal(out.detach().cpu(), ref.detach().to(dtype))
    out.backward(gout.to(DEV))
    assert torch.equal(xd.grad.cpu(), gout)
    tol = 1e-5 if dtype == torch.float32 else 1e-2          # d emb leaves in the weight's dtype
    ref_dw = wr.grad.numpy()
    np.testing.assert_allclose(wd.grad.float().cpu().numpy(), ref_dw, rtol=tol, atol=tol * np.abs(ref_dw).max())


# ------------------------------------------------------------------------------------------------ full-size properties
def test_full_size_properties(ops):
    """configs[1] sizes (B32 x T1500 x d1024, rate 4, 8 heads, L128): too big for the oracle in seconds, so the path is
    checked through size-independent properties -- exact scalings, orderings, counts, row sums, shift invariance."""
    from star_amd.nugget import NuggetScorer
    torch.manual_seed(5)
    B, T, d, h, L, rate = 32, 1500, 1024, 8, 128, 4
    K = int(np.ceil(T / rate))
    rng = np.random.default_rng(3)
    lengths = [T] + [int(v) for v in rng.integers(T // 3, T + 1, size=B - 1)]
    len_dev = torch.tensor(lengths, dtype=torch.int32, device=DEV)
    x = torch.randn(B, T, d, device=DEV)
    x = ((x - x.mean(-1, keepdim=True)) / x.std(-1, keepdim=True)).to(torch.bfloat16)

    # scorer: doubling w2 and b2 doubles every score exactly (a power-of-two scaling commutes with every rounding)
    sc = NuggetScorer(d).to(DEV)
    with torch.no_grad():
        s1 = sc(x).squeeze(-1).float()
        sc[2].weight.mul_(2.0)
        sc[2].bias.mul_(2.0)
        s2 = sc(x).squeeze(-1).float()
    assert torch.equal(s2, 2.0 * s1)

    # select: counts, ordering, type table, idempotence
    nmask, idx, sg, table, pos = ops.topk_select(s1, lengths=len_dev, rate=rate, K=K)
    kb = orc.nugget_counts(np.asarray(lengths), rate)
    np.testing.assert_array_equal(nmask.sum(1).cpu().numpy(), kb)
    np.testing.assert_array_equal(table.sum(1).cpu().numpy(), kb)
    idx_c, nm_c = idx.cpu().numpy(), nmask.cpu().numpy().astype(bool)
    for b in range(B):
        kept = idx_c[b, nm_c[b]]
        assert (np.diff(kept) > 0).all() and kept[-1] == lengths[b] - 1          # ascending, last valid frame kept
        assert (np.diff(idx_c[b, ~nm_c[b]]) > 0).all()                          # pad group ascending too
    assert torch.equal(pos[nmask.bool()], idx[nmask.bool()]) and (pos[~nmask.bool()] == -1).all()
    assert torch.equal(sg, s1.gather(1, idx))
    again = ops.topk_select(s1, lengths=len_dev, rate=rate, K=K)
    assert all(torch.equal(a, b) for a, b in zip((nmask, idx, sg, table, pos), again))

    # gather: rows are copies; the backward of ones counts how often a frame was taken
    s_rep = torch.randn(B, T, d, device=DEV).to(torch.bfloat16).requires_grad_()
    mem = ops.gather_rows(s_rep, idx)
    probe = rng.integers(0, K, size=64)
    for b, j in zip(rng.integers(0, B, size=64), probe):
        assert torch.equal(mem[b, j], s_rep[b, idx[b, j]])
    mem.backward(torch.ones_like(mem))
    counts = torch.zeros(B, T, device=DEV).scatter_add_(1, idx, torch.ones(B, K, device=DEV))
    assert torch.equal(s_rep.grad[..., 0].float(), counts) and torch.equal(s_rep.grad[..., -1].float(), counts)

    # attention core: exact linearity in V, P rows sum to 1, shift invariance in the bias, sum_j dbias = 0
    key_len = torch.from_numpy(kb.astype(np.int32)).to(DEV)
    q = torch.randn(B, h, L, d // h, device=DEV).to(torch.bfloat16)
    kk = mem.detach().view(B, K, h, d // h).permute(0, 2, 1, 3)
    vv = torch.randn(B, K, h, d // h, device=DEV).to(torch.bfloat16).permute(0, 2, 1, 3)
    bias = sg.detach().clone().requires_grad_()
    o1, p1 = ops.xattn(q, kk, vv, bias=bias, key_len=key_len, need_probs=True)
    o2, _ = ops.xattn(q, kk, vv * 2, bias=bias, key_len=key_len)
    assert torch.equal(o2, o1 * 2)
    rows = p1.sum(-1)
    assert (rows - 1).abs().max() < 1e-4
    dead = torch.arange(K, device=DEV)[None, :] >= key_len[:, None]
    assert (p1.permute(0, 3, 1, 2)[dead] == 0).all()                               # masked keys: exact zeros
    o3, _ = ops.xattn(q, kk, vv, bias=bias.detach() + 3.0, key_len=key_len)
    assert (o3.float() - o1.float()).abs().max() <= 2e-2 * o1.float().abs().max()
    o1.backward(torch.randn_like(o1))
    live = ~dead
    tot = (bias.grad * live).sum(1).abs()
    assert (tot <= 2e-3 * (bias.grad * live).abs().sum(1) + 1e-6).all()            # softmax is shift invariant
    assert (bias.grad[dead] == 0).all()

    # fire scan: ascending positions, forced fire at the last valid frame, count == oracle on a few rows
    alphas = torch.rand(B, T, device=DEV) * 0.15
    fidx, fmask, fcounts, mx, _ = ops.cif_fire_scan(alphas, lengths=len_dev, cap=T)
    fi, fc = fidx.cpu().numpy(), fcounts.cpu().numpy()
    for b in range(B):
        f = fi[b, :fc[b]]
        assert (np.diff(f) > 0).all() and (lengths[b] - 1) in f and (fi[b, fc[b]:] == -1).all()
    sub = [0, 7, 31]
    o_mask, o_idx, _ = orc.fire_scan(alphas[sub].cpu().numpy(), lmask([lengths[i] for i in sub], T), with_block_mask=False)
    for r, b in enumerate(sub):
        np.testing.assert_array_equal(fi[b, :fc[b]], o_idx[r, :o_mask[r].sum()])


def test_xattn_split_backward_path_forced(ops):
    """The dispatcher sends almost every short-target shape to the fused backward; the split pair (ring dQ +
    barrier-free dK/dV) stays the path for few (batch, head) pairs with long memories.  Both are forced here through
    the test-only star_test_override() switch (same process) and compared; the general dQ / dK,dV kernels too."""
    res = {}
    modes = {"fused": [(ops.OVR_XA_BWD, 2)], "fused4": [(ops.OVR_XA_BWD, 3)], "split": [(ops.OVR_XA_BWD, 1)],
             "general": [(ops.OVR_XA_DQ, 1), (ops.OVR_XA_DKV, 1)]}
    try:
        for mode, switches in modes.items():
            for w, v in switches:
                ops.test_override(w, v)
            out = {}
            for n, (B, h, L, K, hd, ilk) in enumerate([(6, 8, 100, 300, 128, False), (9, 4, 128, 150, 64, True)]):
                g = torch.Generator().manual_seed(100 + n)
                mk = lambda *sh: torch.randn(*sh, generator=g).to(torch.bfloat16).to(DEV).requires_grad_()
                q, k, v = mk(B, h, L, hd), mk(B, h, K, hd), mk(B, h, K, hd)
                sc = torch.randn(B, K, generator=g).to(DEV).requires_grad_()
                klen = torch.tensor([K] + [max(3, K - 29 * (i + 1)) for i in range(B - 1)], dtype=torch.int32).to(DEV)
                o, _ = ops.xattn(q, k, v, bias=sc, key_len=klen, ilk_diag=2 if ilk else -1)
                o.backward(torch.randn(o.shape, generator=g).to(torch.bfloat16).to(DEV))
                for name, t in (("dq", q.grad), ("dk", k.grad), ("dv", v.grad), ("dsc", sc.grad)):
                    out[f"{name}{n}"] = t.float().cpu().numpy()
            res[mode] = out
            for w, _ in switches:
                ops.test_override(w, 0)
    finally:
        for w in (ops.OVR_XA_BWD, ops.OVR_XA_DQ, ops.OVR_XA_DKV, ops.OVR_XA_NG):
            ops.test_override(w, 0)
    for key in res["fused"]:
        for other in ("fused4", "split", "general"):
            a, b = res["fused"][key], res[other][key]
            assert np.abs(a - b).max() <= 2e-2 * np.abs(b).max() + 1e-6, (key, other)
