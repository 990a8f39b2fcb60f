"""Randomised parity sweep: random shapes / lengths / switches for every kernel family against the CPU oracle.

The parametrised tests pin the shapes that matter (the configs, the kernels' dispatch boundaries); this file walks the
space between them.  STAR_FUZZ_CASES cases per family (default 24, so the suite stays short; a sweep is
`STAR_FUZZ_CASES=300 STAR_FUZZ_SEED=... pytest tests/test_gpu_fuzz.py`), every case seeded by (STAR_FUZZ_SEED, family,
case number) and named in the failure message, so that a failing case can be replayed alone with STAR_FUZZ_ONLY=<n>.
Bars as in test_gpu_parity.py: integer work bit-exact, bf16-MFMA work 2e-2 of the tensor's scale against the fp32
oracle on the same bf16-rounded operands.
"""
import os

import numpy as np
import pytest
import torch

from oracle import nugget_oracle as orc
from tests.test_oracle_golden import lmask

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
CASES = int(os.environ.get("STAR_FUZZ_CASES", "24"))
SEED = int(os.environ.get("STAR_FUZZ_SEED", "20261111"))
ONLY = os.environ.get("STAR_FUZZ_ONLY")
BIG = bool(os.environ.get("STAR_FUZZ_BIG"))
PROGRESS = os.environ.get("STAR_FUZZ_PROGRESS")          # a file that long sweeps append a line to every 200 cases


@pytest.fixture(scope="module")
def ops():
    from star_amd import ops as o
    return o


def _cases(family):
    fam = sum(ord(c) * (i + 1) for i, c in enumerate(family))
    for n in range(CASES):
        if ONLY is not None and int(ONLY) != n:
            continue
        if n and n % (10 if BIG else 200) == 0 and PROGRESS:          # a long sweep shows that it is alive
            with open(PROGRESS, "a") as f:
                f.write(f"fuzz {family}: {n}/{CASES}\n")
        yield n, np.random.default_rng([SEED, fam, n])


def bf16_round(t):
    return t.to(torch.bfloat16).to(torch.float32)


def _close(a, b, what, tol=2e-2, floor=0.0):
    a = a.detach().float().cpu().numpy()
    b = b.detach().float().cpu().numpy()
    scale = max(np.abs(b).max(), floor) if b.size else 0.0
    err = np.abs(a - b).max() if a.size else 0.0
    assert err <= tol * scale + 1e-6, f"{what}: max err {err} scale {scale}"


def _tg(rng):
    return torch.Generator().manual_seed(int(rng.integers(1 << 31)))


# ------------------------------------------------------------------------------------------------ attention
def test_fuzz_xattn(ops):
    for n, rng in _cases("xattn"):
        hd = int(rng.choice([64, 96, 128]))
        h = int(rng.choice([1, 2, 3, 4, 8]))
        B = int(rng.choice([1, 2, 3, 5, 24, 26, 33]))        # B*h >= 192 reaches the fused backward
        L = int(rng.choice([1, 7, 16, 37, 64, 100, 128, 129, 200, 257, 300]))
        K = int(rng.choice([1, 2, 15, 16, 17, 63, 64, 65, 127, 128, 129, 200, 375, 400, 750]))
        lim = 3e8                                           # keeps the fp32 oracle to a second or so
        if BIG:            # STAR_FUZZ_BIG=1: chip-filling shapes (B*h >= 192: fused / ring kernels), seconds of oracle each
            h, B, lim = 8, int(rng.choice([24, 26, 32, 33, 40])), 4e9
            L = int(rng.choice([37, 64, 100, 128, 129, 200, 227, 257]))
            K = int(rng.choice([127, 128, 129, 200, 229, 375, 400, 750]))
        if B * h * L * K * hd > lim:
            B = max(1, int(lim // (h * L * K * hd)))
        ilk = bool(rng.integers(2))
        p = float(rng.choice([0.0, 0.0, 0.1, 0.3]))
        bias_mode = int(rng.integers(3))                     # 0: [B,K] logits, 1: none, 2: prefix bias shorter than K
        klen = rng.integers(1, K + 1, size=B)
        klen[0] = K
        what = f"xattn case {n}: B{B} h{h} L{L} K{K} hd{hd} ilk{ilk} p{p} bias{bias_mode} klen{klen.tolist()[:4]}"
        g = _tg(rng)
        q = torch.randn(B, h, L, hd, generator=g)
        k = torch.randn(B, h, K, hd, generator=g)
        v = torch.randn(B, h, K, hd, generator=g)
        gout = torch.randn(B, h, L, hd, generator=g)
        Kb = K if bias_mode == 0 else (0 if bias_mode == 1 else int(rng.integers(1, K + 1)))
        sc = torch.randn(B, Kb, generator=g) if Kb else None
        sc_full = torch.zeros(B, K)
        if Kb:
            sc_full[:, :Kb] = sc
        qr, kr, vr = (bf16_round(t).requires_grad_() for t in (q, k, v))
        scr = sc_full.clone().requires_grad_()
        kmask = torch.from_numpy(lmask(klen, K))
        kw = {}
        seed, offset = int(rng.integers(1 << 62)), int(rng.integers(1 << 20))
        if p > 0:
            keep, p_eff = orc.dropout_keep_mask(seed, offset, B, h, L, K, p)
            kw = dict(drop_mask=torch.from_numpy(keep), drop_p=p_eff)
        o_ref, p_ref = orc.xattn_core(qr, kr, vr, kmask, scr, ilk, **kw)
        (o_ref * bf16_round(gout)).sum().backward()
        qd, kd, vd = (t.to(torch.bfloat16).to(DEV).requires_grad_() for t in (q, k, v))
        scd = sc.to(DEV).requires_grad_() if Kb else None
        key_len = torch.tensor(klen, dtype=torch.int32, device=DEV)
        dkw = dict(dropout_p=p, dropout_rng=(seed, offset)) if p > 0 else {}
        o, pw = ops.xattn(qd, kd, vd, bias=scd, key_len=key_len, ilk_diag=2 if ilk else -1, need_probs=True, **dkw)
        o.backward(gout.to(torch.bfloat16).to(DEV))
        # queries that see no key at all cannot occur: key 0 is always valid and visible (ilk: key <= query + 2)
        _close(pw, p_ref, what + " P", 2e-3)
        _close(o, o_ref, what + " O")
        # dS = P (dP - delta) with delta = rowsum(dO * O) taken from the bf16 O, as every flash-style backward does:
        # where the exact dS cancels to ~0 (one visible key: P = 1, dP = delta) the rounding of O (2^-9 |dO . V|,
        # |dO . V| ~ sqrt(hd)) is what is left, so the scale has a floor of that natural magnitude (times the root of
        # the number of dS terms a result sums: L queries for dK, L * h for dscores)
        fl = hd ** 0.5
        _close(qd.grad, qr.grad, what + " dQ", floor=fl)
        _close(kd.grad, kr.grad, what + " dK", floor=fl * L ** 0.5)
        _close(vd.grad, vr.grad, what + " dV")
        if Kb:
            _close(scd.grad, scr.grad[:, :Kb], what + " dscores", floor=fl * (L * h) ** 0.5)
        assert torch.isfinite(o.float()).all() and torch.isfinite(kd.grad.float()).all(), what


# ------------------------------------------------------------------------------------------------ select
def test_fuzz_select(ops):
    for n, rng in _cases("select"):
        B = int(rng.integers(1, 40))
        T = int(rng.choice([1, 2, 5, 63, 64, 255, 256, 257, 1000, 1500, 3000, 8191]))
        rate = float(rng.choice([1, 2, 4, 8, 12, 3.5]))
        scores = rng.standard_normal((B, T)).astype(np.float32)
        mode = int(rng.integers(4))
        if mode == 1:
            scores = np.round(scores * 2) / 2                  # heavy ties
        elif mode == 2:
            scores[:] = scores[0, 0]                           # all equal
        elif mode == 3:
            scores[rng.random((B, T)) < 0.1] = -0.0            # signed zeros tie with +0.0
            scores[rng.random((B, T)) < 0.1] = 0.0
        lengths = rng.integers(1, T + 1, size=B)
        if rng.integers(2):
            lengths[0] = T
        what = f"select case {n}: B{B} T{T} rate{rate} mode{mode}"
        mask = lmask(lengths, T)
        use_mask = bool(rng.integers(2))
        kw = dict(attn_mask=torch.from_numpy(mask).to(DEV)) if use_mask else \
            dict(lengths=torch.from_numpy(lengths).to(torch.int32).to(DEV))
        nmask, idx, sg, table, pos, rows = ops.topk_select(torch.from_numpy(scores).to(DEV), rate=rate, want_rows=True,
                                                           **kw)
        o_mask, o_idx, o_sg = orc.topk_select(scores, mask, rate)
        o_table, o_pos = orc.type_table(T, o_idx, o_mask)
        for got, ref, nm in ((nmask, o_mask, "mask"), (idx, o_idx, "idx"), (sg, o_sg, "scores"), (table, o_table, "table"),
                             (pos, o_pos, "pos")):
            assert np.array_equal(got.cpu().numpy(), ref), f"{what}: {nm}"
        assert np.array_equal(rows.cpu().numpy(), (o_idx + np.arange(B)[:, None] * T).reshape(-1)), what


# ------------------------------------------------------------------------------------------------ scorer
def test_fuzz_scorer(ops):
    for n, rng in _cases("scorer"):
        d = int(rng.choice([128, 256, 384, 768, 1024]))
        B = int(rng.integers(1, 9))
        T = int(rng.choice([1, 17, 64, 255, 256, 257, 500, 700, 1100]))
        if B * T * d * d > 6e9:
            B = max(1, int(6e9 // (T * d * d)))
        form = int(rng.integers(3))                            # 0: dense, 1: ragged tiles, 2: compact
        lengths = rng.integers(0, T + 1, size=B)
        lengths[int(rng.integers(B))] = T
        if form == 0:
            lengths[:] = T
        hinted = bool(rng.integers(2))
        nt = int(rng.integers(2))                              # force the NT dW1 GEMM in half of the cases
        what = f"scorer case {n}: B{B} T{T} d{d} form{form} hinted{hinted} nt{nt} lengths{lengths.tolist()}"
        g = _tg(rng)
        lim = 1 / d ** 0.5
        x = torch.randn(B, T, d, generator=g)
        w1 = (torch.rand(d, d, generator=g) * 2 - 1) * lim
        b1 = (torch.rand(d, generator=g) * 2 - 1) * lim
        w2 = (torch.rand(1, d, generator=g) * 2 - 1) * lim
        b2 = (torch.rand(1, generator=g) * 2 - 1) * lim
        mask = torch.from_numpy(lmask(lengths, T))
        gs = torch.randn(B, T, generator=g) * mask
        keep = mask.clone()
        if hinted:
            keep = keep & (torch.rand(B, T, generator=g) < 0.3)
            if not keep.any():
                keep[int(np.argmax(lengths)), 0] = True
            gs = gs * keep
        xr, w1r = bf16_round(x).requires_grad_(), bf16_round(w1).requires_grad_()
        b1r, w2r, b2r = b1.clone().requires_grad_(), w2.clone().requires_grad_(), b2.clone().requires_grad_()
        s_ref = orc.scorer(xr, w1r, b1r, w2r, b2r)
        (s_ref * gs).sum().backward()
        xd = x.to(torch.bfloat16).to(DEV).requires_grad_()
        params = [t.to(DEV).requires_grad_() for t in (w1, b1, w2, b2)]
        valid = None if form == 0 else ops.ValidFrames(lengths.tolist(), T, DEV, compact=(form == 2))
        hint = ops.ActiveRows()
        try:
            ops.test_override(ops.OVR_SCORER_NT, nt)
            s = ops.scorer(xd, *params, hint, valid)
            sv = s.detach().cpu().reshape(B, T)
            np.testing.assert_allclose(sv[mask].numpy(), s_ref.detach().reshape(B, T)[mask].numpy(), rtol=2e-3, atol=2e-3,
                                       err_msg=what)
            if form:
                assert (sv[~mask] == 0).all(), what + ": pad scores"
            if hinted:
                # the hint may list frames without a gradient as well (pad slots of the select): add a few
                extra = (torch.rand(B, T, generator=g) < 0.05) & ~keep
                rows = torch.nonzero((keep | extra).reshape(-1)).flatten().to(torch.int32).to(DEV)
                hint.set(rows)
            (s.reshape(B, T) * gs.to(DEV)).sum().backward()
        finally:
            ops.test_override(ops.OVR_SCORER_NT, 0)
        # relu'(H) is a step: a pre-activation within summation-order noise of 0 (|H| < 2e-5; the kernel and the oracle
        # add the same d products in different orders) may land on either side, and its term ds w2[n] x[r][:] is then in
        # one gradient and not in the other.  Such (frame, unit) pairs get exactly their term as extra allowance.
        with torch.no_grad():
            hpre = xr.reshape(-1, d) @ w1r.t() + b1r
            amb = ((hpre.abs() < 2e-5) & (gs.reshape(-1, 1) != 0)).float()
            gabs = gs.reshape(-1, 1).abs()
            allow = {"dW1": (amb * gabs).t() @ xr.reshape(-1, d).abs() * w2r.abs().reshape(-1, 1),
                     "db1": (amb * gabs).sum(0) * w2r.abs().reshape(-1),
                     "dX": ((amb * gabs * w2r.abs().reshape(1, -1)) @ w1r.abs()).reshape(B, T, d)}
        for got, ref, nm in zip([p.grad for p in params] + [xd.grad], (w1r.grad, b1r.grad, w2r.grad, b2r.grad, xr.grad),
                                ("dW1", "db1", "dw2", "db2", "dX")):
            got = got.detach().float().cpu().reshape(ref.shape)
            err = (got - ref).abs() - allow.get(nm, torch.zeros(())).reshape(ref.shape if nm in allow else ())
            lim = 2e-2 * float(ref.abs().max()) + 1e-6
            assert float(err.max()) <= lim, (f"{what} {nm}: max err {float(err.max())} over the allowance, limit {lim}, "
                                             f"{int(amb.sum())} ambiguous pre-activations")
        assert (xd.grad.float().cpu()[~keep] == 0).all(), what + ": dX rows without a score gradient"


# ------------------------------------------------------------------------------------------------ gather
def test_fuzz_gather(ops):
    for n, rng in _cases("gather"):
        dtype = [torch.bfloat16, torch.float32][int(rng.integers(2))]
        B = int(rng.integers(1, 12))
        T = int(rng.choice([1, 3, 64, 100, 333, 1500]))
        K = int(rng.choice([1, 2, 9, 64, 100, 375, 500]))
        d = int(rng.choice([8, 96, 128, 768, 1024]))
        what = f"gather case {n}: B{B} T{T} K{K} d{d} {dtype}"
        g = _tg(rng)
        src = torch.randn(B, T, d, generator=g).to(dtype)
        idx = torch.randint(0, T, (B, K), generator=g).sort(dim=1).values
        npad = int(rng.integers(0, K))                          # simul-style -1 pads at the end of some rows
        if npad:
            idx[int(rng.integers(B)), K - npad:] = -1
        s_dev = src.to(DEV).requires_grad_()
        out = ops.gather_rows(s_dev, idx.to(DEV))
        assert torch.equal(out.cpu(), orc.gather_rows(src, idx)), what
        gout = torch.randn(B, K, d, generator=g).to(dtype)
        out.backward(gout.to(DEV))
        acc = torch.zeros(B, T, d, dtype=torch.float32)
        acc.scatter_add_(1, torch.where(idx < 0, idx + T, idx)[:, :, None].expand(B, K, d), gout.float())
        # duplicates are summed in fp32 in slot order and rounded once; an fp32 scatter_add may differ in the last bit
        ref = acc.to(dtype).float()
        tol = 1e-6 if dtype == torch.float32 else 8e-3
        np.testing.assert_allclose(s_dev.grad.float().cpu().numpy(), ref.numpy(), rtol=tol, atol=tol * float(ref.abs().max()),
                                   err_msg=what)


# ------------------------------------------------------------------------------------------------ CIF
def test_fuzz_cif(ops):
    for n, rng in _cases("cif"):
        B = int(rng.integers(1, 70))
        T = int(rng.choice([1, 2, 65, 333, 1500, 3000]))
        lengths = rng.integers(1, T + 1, size=B)
        if rng.integers(2):
            lengths[0] = T
        what = f"cif case {n}: B{B} T{T}"
        mask = lmask(lengths, T)
        alphas = (rng.random((B, T)) * float(rng.choice([0.05, 0.3, 0.9]))).astype(np.float32)
        if T > 6 and rng.integers(2):
            alphas[int(rng.integers(B)), 5] = 3.7                 # carries through several frames
        o_mask, o_idx, _ = orc.fire_scan(alphas, mask, with_block_mask=False)
        scores = rng.standard_normal((B, T)).astype(np.float32)
        idx, nmask, counts, mx, sg = ops.cif_fire_scan(torch.from_numpy(alphas).to(DEV),
                                                       lengths=torch.from_numpy(lengths).to(torch.int32).to(DEV),
                                                       scores=torch.from_numpy(scores).to(DEV))
        kf = int(mx.item())
        assert kf == o_idx.shape[1], what
        assert np.array_equal(idx[:, :kf].cpu().numpy(), o_idx), what
        assert np.array_equal(nmask[:, :kf].cpu().numpy(), o_mask), what
        assert np.array_equal(sg[:, :kf].cpu().numpy(), scores[np.arange(B)[:, None], o_idx]), what
        # resize + quantity loss on sigmoid(scores)
        tl = np.maximum(1, (lengths * float(rng.choice([0.05, 0.25, 0.9]))).astype(np.int64))
        a = (1 / (1 + np.exp(-scores))).astype(np.float32) * mask
        ref = orc.resize(torch.from_numpy(a), torch.from_numpy(tl)).numpy()
        out = ops.cif_resize(torch.from_numpy(a).to(DEV), torch.from_numpy(tl).to(torch.int32).to(DEV)).cpu().numpy()
        np.testing.assert_allclose(out, ref, rtol=5e-5, atol=1e-7, err_msg=what + " resize")


# ------------------------------------------------------------------------------------------------ 8f rows
def test_fuzz_cif_aggregate(ops):
    for n, rng in _cases("cifagg"):
        dtype = [torch.bfloat16, torch.float32][int(rng.integers(2))]
        B = int(rng.integers(1, 20))
        T = int(rng.choice([1, 2, 65, 129, 300, 700]))
        d = int(rng.choice([8, 64, 100, 768, 1024]))
        lengths = rng.integers(1, T + 1, size=B)
        what = f"cif aggregate case {n}: B{B} T{T} d{d} {dtype}"
        mask = lmask(lengths, T)
        alphas_np = (rng.random((B, T)).astype(np.float32) * float(rng.choice([0.1, 0.4, 0.95]))) * mask
        if rng.integers(2):
            alphas_np[int(rng.integers(B))] *= 0.001                  # a row that never fires
        s_np = rng.standard_normal((B, T, d)).astype(np.float32)
        s_ref = torch.from_numpy(s_np).to(dtype).float().requires_grad_()
        a_ref = torch.from_numpy(alphas_np).clone().requires_grad_()
        o_cif, o_mask, o_pos = orc.cif_aggregate(a_ref, s_ref)
        a = torch.from_numpy(alphas_np).to(DEV).requires_grad_()
        s = torch.from_numpy(s_np).to(dtype).to(DEV).requires_grad_()
        cif, cmask, idx, counts, mx = ops.cif_aggregate(a, s, cap=T)
        kf = o_pos.shape[1]
        assert int(mx.item()) == kf, what
        assert np.array_equal(idx[:, :kf].cpu().numpy(), o_pos.numpy()), what
        assert (idx[:, kf:] == -1).all() and (cif[:, kf:] == 0).all(), what
        assert np.array_equal(cmask[:, :kf].cpu().numpy(), o_mask.numpy().astype(np.uint8)), what
        bf = dtype == torch.bfloat16
        _close(cif[:, :kf], o_cif, what + " cif", 1e-2 if bf else 1e-5)
        if kf == 0:
            continue
        gout = torch.from_numpy(rng.standard_normal((B, kf, d)).astype(np.float32)).to(dtype)
        (o_cif * gout.float()).sum().backward()
        gfull = torch.zeros(B, T, d, dtype=dtype)
        gfull[:, :kf] = gout
        cif.backward(gfull.to(DEV))
        _close(s.grad, s_ref.grad, what + " d states", 2e-2 if bf else 1e-4)
        _close(a.grad, a_ref.grad, what + " d alphas", 2e-2 if bf else 2e-4)


def test_fuzz_type_embed_and_gather_layernorm(ops):
    from star_amd.nugget import gather_nuggets_layernorm
    for n, rng in _cases("typeln"):
        dtype = [torch.bfloat16, torch.float32][int(rng.integers(2))]
        B = int(rng.integers(1, 12))
        T = int(rng.choice([1, 3, 77, 130, 700]))
        K = int(rng.choice([1, 5, 40, 90, 375]))
        d = int(rng.choice([8, 200, 520, 768, 1024, 1536]))
        what = f"type embed / gather LN case {n}: B{B} T{T} K{K} d{d} {dtype}"
        g = _tg(rng)
        bf = dtype == torch.bfloat16
        # x + type_embed(table)
        x = torch.randn(B, T, d, generator=g).to(dtype)
        w = (torch.randn(2, d, generator=g) * 0.1).to(dtype)
        table = (torch.rand(B, T, generator=g) < float(rng.choice([0.0, 0.25, 1.0]))).long()
        gout = torch.randn(B, T, d, generator=g).to(dtype)
        xr, wr = x.float().clone().requires_grad_(), w.float().clone().requires_grad_()
        ref = orc.type_embed_add(xr, table, wr)
        (ref * gout.float()).sum().backward()
        xd, wd = x.detach().to(DEV).requires_grad_(), w.detach().to(DEV).requires_grad_()
        out = ops.type_embed_add(xd, table.to(DEV), wd)
        assert torch.equal(out.detach().cpu(), ref.detach().to(dtype)), what
        out.backward(gout.to(DEV))
        assert torch.equal(xd.grad.cpu(), gout), what
        _close(wd.grad, wr.grad, what + " d emb", 1e-2 if bf else 1e-5)
        # LayerNorm of the gathered rows only
        idx = np.sort(rng.integers(0, T, size=(B, K)), axis=1)
        npad = int(rng.integers(0, K))
        if npad:
            idx[:, K - npad:] = -1
        h32 = (torch.randn(B, T, d, generator=g) * 2 + 0.5).to(dtype).float()
        lw, lb = 1 + 0.3 * torch.randn(d, generator=g), 0.2 * torch.randn(d, generator=g)
        gout32 = torch.randn(B, K, d, generator=g).to(dtype).float()
        hr, wr2, br2 = h32.clone().requires_grad_(), lw.clone().requires_grad_(), lb.clone().requires_grad_()
        ref = orc.gather_layernorm(hr, torch.from_numpy(idx), wr2, br2, 1e-5)
        (ref * gout32).sum().backward()
        lnd = torch.nn.LayerNorm(d, eps=1e-5).to(DEV)
        with torch.no_grad():
            lnd.weight.copy_(lw)
            lnd.bias.copy_(lb)
        hdv = h32.to(dtype).to(DEV).requires_grad_()
        o2 = gather_nuggets_layernorm(hdv, torch.from_numpy(idx).to(DEV), lnd)
        (o2.float() * gout32.to(DEV)).sum().backward()
        tol = 1e-2 if bf else 1e-5
        _close(o2, ref, what + " LN out", tol)
        _close(hdv.grad, hr.grad, what + " LN d hidden", 2 * tol)
        _close(lnd.weight.grad, wr2.grad, what + " LN d weight", 2 * tol if bf else 1e-4)
        _close(lnd.bias.grad, br2.grad, what + " LN d bias", 2 * tol if bf else 1e-4)


# ------------------------------------------------------------------------------------------------ whole steps
# ------------------------------------------------------------------------------------------------ streaming push
def test_fuzz_stream_push(ops):
    """star_cif_stream_push over random batches, lengths, logit scales (saturated sigmoids and exact zeros included),
    row widths, dtypes, chunkings and end-of-utterance timings against the offline kernels: memory rows, logits,
    positions and counts bit for bit (the offline fire scan itself is pinned to the reference's goldens elsewhere)."""
    from tests.test_gpu_round3 import _offline_memory, _check_memory
    for n, rng in _cases("stream"):
        B = int(rng.integers(1, 7))
        T = int(rng.choice([1, 2, 15, 16, 17, 63, 64, 65, 200, 333, 700]))
        dtype = torch.bfloat16 if rng.random() < 0.7 else torch.float32
        d = int(rng.choice([8, 64, 136, 256])) if dtype == torch.bfloat16 else int(rng.choice([4, 36, 64]))
        g = _tg(rng)
        scores = torch.randn(B, T, generator=g) * float(rng.choice([0.5, 1.5, 6.0])) + float(rng.choice([-3.0, -1.0, 1.0]))
        if rng.random() < 0.3:
            scores[int(rng.integers(B))] = 25.0                       # sigmoid == 1.0 exactly: the generic chain
        if rng.random() < 0.3 and T > 4:
            a = int(rng.integers(0, T - 2))
            scores[int(rng.integers(B)), a:a + int(rng.integers(1, T - a))] = -200.0      # exact zero weights
        scores = scores.to(DEV)
        s = torch.randn(B, T, d, generator=g).to(dtype).to(DEV)
        lengths = [int(x) for x in rng.integers(1, T + 1, B)]
        lengths[int(rng.integers(B))] = T
        idx, counts, sg, mem_off = _offline_memory(ops, scores, s, lengths)
        kmax = int(counts.max()) + int(rng.integers(0, 3))
        ln = torch.tensor(lengths, dtype=torch.int32, device=DEV)
        unknown = torch.full_like(ln, 2 ** 31 - 1)
        sizes, left = [], T
        while left:
            c = int(min(left, rng.integers(1, max(2, int(rng.choice([4, 40, 200]))))))
            sizes.append(c)
            left -= c
        mode = str(rng.choice(["known", "with_last_frame", "late"]))
        st = ops.StreamState(B, d, max(kmax, 1), DEV, dtype)
        off = 0
        for c in sizes:
            cur = ln if mode == "known" else torch.where((ln <= off + c) if mode == "with_last_frame" else (ln < off + c),
                                                         ln, unknown)
            ops.cif_stream_push(scores[:, off:off + c].contiguous(), s[:, off:off + c].contiguous(), cur.contiguous(), st)
            off += c
        if mode == "late":
            ops.cif_stream_push(None, None, ln, st)
        assert (st.pos == T).all()
        _check_memory(st, idx, counts, sg, mem_off, f"stream case {n} (B{B} T{T} d{d} {dtype} {mode} {sizes[:6]})")


# ------------------------------------------------------------------------------------------------ projections
def test_fuzz_projection_gemm_and_colsum(ops):
    """star_gemm_nt_bf16 (out = A . B^T + bias, one bf16 rounding) and star_colsum over random shapes, leading
    dimensions and tile-edge cases against fp32 torch on the same bf16 operands."""
    for n, rng in _cases("proj"):
        M = int(rng.choice([1, 17, 255, 256, 257, 1000, 4096, 5000]))
        N = int(rng.choice([64, 128, 192, 256, 320, 1024, 2048]))
        K = int(rng.choice([64, 128, 256, 576, 1024]))
        g = _tg(rng)
        lda = K + int(rng.choice([0, 0, 64]))
        a_full = torch.randn(M, lda, generator=g).to(torch.bfloat16).to(DEV)
        a = a_full[:, :K]
        b = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16).to(DEV)
        bias = torch.randn(N, generator=g).to(DEV) if rng.random() < 0.7 else None
        out = ops.gemm_nt_bf16(a, b, bias)
        ref = a.float() @ b.float().t() + (bias if bias is not None else 0.0)
        _close(out, ref, f"proj case {n} (M{M} N{N} K{K} lda{lda} bias={bias is not None})", tol=1e-2)
        for dt in (torch.bfloat16, torch.float32):
            x = torch.randn(M, N, generator=g).to(dt).to(DEV)
            cs = ops.colsum(x)
            want = x.double().sum(0)
            err = (cs.double() - want).abs().max().item()
            assert err <= 1e-5 * max(x.float().abs().sum(0).max().item(), 1.0), f"colsum case {n} ({M}x{N} {dt}): {err}"


def test_fuzz_tn_batched_and_transposes(ops):
    """star_gemm_tn_batched_bf16 ([x_i^T g_i for i], fp32), star_colsum_batched and star_transpose_list_bf16 over random
    batch sizes, r-range counts, shapes and row strides against fp64 / torch on the same operands."""
    for n, rng in _cases("tnb"):
        ks = int(rng.choice([1, 1, 2, 4]))
        nb = int(rng.integers(1, 16 // ks + 1))
        R = 64 * ks * int(rng.integers(1, 9))
        M = int(rng.choice([128, 136, 256, 264, 512, 1024]))
        N = int(rng.choice([128, 200, 256, 384, 1024]))
        pa, pb = int(rng.choice([0, 0, 8, 64])), int(rng.choice([0, 0, 8, 64]))
        g = _tg(rng)
        xs = [torch.randn(R, M + pa, generator=g).to(torch.bfloat16).to(DEV)[:, :M] for _ in range(nb)]
        gs = [torch.randn(R, N + pb, generator=g).to(torch.bfloat16).to(DEV)[:, :N] for _ in range(nb)]
        out = ops.gemm_tn_batched(xs, gs, ks)
        what = f"tnb case {n} (nb{nb} ks{ks} R{R} M{M} N{N} pads {pa},{pb})"
        for i in range(nb):
            ref = xs[i].double().t() @ gs[i].double()
            err = (out[i].double() - ref).abs().max().item()
            assert err <= 1e-4 * max(ref.abs().max().item(), 1.0), f"{what} product {i}: {err}"
        cs = ops.colsum_batched(gs)
        for i in range(nb):
            assert torch.equal(cs[i], ops.colsum(gs[i])), f"{what} colsum {i}"
        dt = torch.bfloat16 if rng.random() < 0.5 else torch.float32
        rows, cols = 64 * int(rng.integers(1, 6)), 64 * int(rng.integers(1, 6))
        ws = [torch.randn(rows, cols, generator=g).to(dt).to(DEV) for _ in range(nb)]
        want = torch.cat([w.to(torch.bfloat16) for w in ws], dim=1).t().contiguous()
        assert torch.equal(ops.transpose_cat_bf16(ws), want), f"{what} transpose {rows}x{cols} {dt}"


def _step_close(a, b, what, tol=3e-2, floor=0.0):
    a, b = a.detach().float().cpu().numpy(), b.detach().numpy()
    scale = max(np.abs(b).max(), floor) + 1e-12
    assert np.abs(a - b).max() <= tol * scale, f"{what}: {np.abs(a - b).max()} vs scale {scale}"


def test_fuzz_nugget_steps(ops):
    """Both NuggetSteps (star_amd/step.py: scorer -> select | CIF -> type table -> gather -> nl x attention, forward and
    backward through the Python mirror's autograd plumbing) against the oracle's fp32 steps on random small shapes."""
    from star_amd.nugget import NuggetScorer
    from star_amd.step import make_inputs, nugget_step_asr, nugget_step_simul
    cpu = lambda t: t.detach().float().cpu()
    for n, rng in _cases("steps"):
        if n % 4:                                   # a step costs a few oracle seconds: a quarter of the case count
            continue
        torch.manual_seed(int(rng.integers(1 << 31)))
        B = int(rng.integers(1, 6))
        T = int(rng.choice([40, 129, 250, 300]))
        d = int(rng.choice([128, 256, 768]))
        h = int(rng.choice({128: [1, 2], 256: [2, 4], 768: [6, 8, 12]}[d]))       # head dim 64, 96 or 128
        nl = int(rng.integers(1, 4))
        lengths = sorted((int(v) for v in rng.integers(max(8, T // 5), T + 1, size=B)), reverse=True)
        lengths[0] = T
        simul = bool(rng.integers(2))
        # loss = sum_l <O_l, G_l> is a sum of B * L * d * nl signed terms that largely cancel; each bf16 O carries 2^-9
        # of relative rounding, so the absolute error grows with the root of the term count
        loss_atol = lambda L: 1e-2 + 2e-4 * (B * L * d * nl) ** 0.5
        scorer = NuggetScorer(d).to(DEV)
        mask = lmask(lengths, T)
        w1, b1, w2, b2 = (cpu(p).requires_grad_() for p in (scorer[0].weight, scorer[0].bias, scorer[2].weight,
                                                            scorer[2].bias))
        w1r = bf16_round(w1.detach()).requires_grad_()
        if simul:
            ilk = bool(rng.integers(2))
            tl = [int(round(0.075 * v)) + 2 for v in lengths]
            L = max(tl)
            what = f"simul step case {n}: lengths{lengths} T{T} d{d} h{h} nl{nl} ilk{ilk}"
            with torch.no_grad():
                scorer[2].bias.fill_(-2.0)
                b2 = cpu(scorer[2].bias).requires_grad_()
            inp = make_inputs(lengths, T, d, h, L, nl, DEV, target_lengths=tl)
            out = nugget_step_simul(scorer, inp, use_ilk=ilk, with_loss=True)
            o_mask, o_idx, _ = orc.fire_scan(out["resized"].cpu().numpy(), mask, with_block_mask=False)
            kf = o_idx.shape[1]
            assert np.array_equal(out["indices"][:, :kf].cpu().numpy(), o_idx), what
            assert (out["indices"][:, kf:] == -1).all(), what
            xr, sr = cpu(inp.x).requires_grad_(), cpu(inp.s).requires_grad_()
            qs = [cpu(q).requires_grad_() for q in inp.qs]
            fires = (out["nugget_mask"].cpu().numpy().astype(bool), out["indices"].cpu().numpy())
            ref = orc.nugget_step_simul(xr, sr, lengths, tl, ilk, w1r, b1, w2, b2, qs,
                                        [(cpu(a), cpu(b)) for a, b in inp.kv_scales], [cpu(g) for g in inp.gs], fires=fires)
            np.testing.assert_allclose(out["scores"].detach().cpu().numpy()[mask], ref["scores"].detach().numpy()[mask],
                                       rtol=2e-3, atol=2e-3, err_msg=what)
            ref["loss"].backward()
            assert abs(out["loss"].item() - ref["loss"].item()) <= 2e-2 * abs(ref["loss"].item()) + loss_atol(L), what
        else:
            rate = float(rng.choice([2, 4, 8]))
            L = int(rng.choice([8, 24, 64, 128]))
            what = f"asr step case {n}: lengths{lengths} T{T} d{d} h{h} nl{nl} L{L} rate{rate}"
            inp = make_inputs(lengths, T, d, h, L, nl, DEV)
            out = nugget_step_asr(scorer, inp, rate, with_loss=True)
            scores = out["scores"].detach().cpu().numpy()
            o_mask, o_idx, _ = orc.topk_select(scores, mask, rate)
            assert np.array_equal(out["indices"].cpu().numpy(), o_idx), what
            assert np.array_equal(out["nugget_mask"].cpu().numpy(), o_mask), what
            xr, sr = cpu(inp.x).requires_grad_(), cpu(inp.s).requires_grad_()
            qs = [cpu(q).requires_grad_() for q in inp.qs]
            # the oracle step runs on the kernel's selection (bit-exact above): a near-tie ranked the other way by the
            # fp32 scores must not skip the comparison of everything downstream
            ref = orc.nugget_step_asr(xr, sr, lengths, rate, w1r, b1, w2, b2, qs,
                                      [(cpu(a), cpu(b)) for a, b in inp.kv_scales], [cpu(g) for g in inp.gs],
                                      select=(o_mask, o_idx))
            np.testing.assert_allclose(scores[mask], ref["scores"].detach().numpy()[mask], rtol=2e-3, atol=2e-3, err_msg=what)
            assert (scores[~mask] == 0).all(), what
            ref["loss"].backward()
            assert abs(out["loss"].item() - ref["loss"].item()) <= 2e-2 * abs(ref["loss"].item()) + loss_atol(L), what
        with torch.no_grad():
            amb = int(((xr.reshape(-1, d) @ w1r.t() + b1).abs() < 2e-5).sum())
        if amb == 0:                                # a ReLU on the fence: see test_fuzz_scorer
            _step_close(scorer[0].weight.grad, w1r.grad, what + " dW1")
            _step_close(scorer[0].bias.grad, b1.grad, what + " db1")
            if not simul:
                _step_close(inp.x.grad, xr.grad, what + " dX")
        # dw2[n] = sum_r G'[r][n] (H[r][n] - b1[n]) from the bf16 G' (2^-9 per term): on a 40-frame utterance the terms do
        # not average out against a max |dw2| that itself comes from cancelling terms
        _step_close(scorer[2].weight.grad, w2.grad, what + " dw2", tol=6e-2)
        if simul:      # the select step's true db2 is 0 (the softmax is invariant under a common shift of the logits)
            # db2 = sum_r dscore[r] is one cancelling sum: measured on the scale of its sibling dw2[n] = sum_r dscore[r] h[r][n]
            _step_close(scorer[2].bias.grad, b2.grad, what + " db2", floor=float(w2.grad.abs().max()))
        _step_close(inp.s.grad, sr.grad, what + " dS")
        for l in range(nl):
            _step_close(inp.qs[l].grad, qs[l].grad, what + f" dQ{l}")
