"""The CPU oracle (oracle/nugget_oracle.py) against the reference-generated golden vectors."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import nugget_oracle as orc

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return dict(np.load(os.path.join(G, name + ".npz")))


def lmask(lengths, t):
    return np.arange(t)[None, :] < np.asarray(lengths)[:, None]


SELECT = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(G, "select_*.npz"))
                if "counts" not in p)
CIF = ["cif_plain", "cif_fixup", "cif_ragged"]
XATTN = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(G, "xattn_*.npz")))


def test_counts_sweep():
    g = load("select_counts")
    for key in g:
        if key == "n":
            continue
        rate = int(key[1:])
        np.testing.assert_array_equal(orc.nugget_counts(g["n"], rate), g[key])


def select_keys(scores, lengths):
    """Selection keys as the reference builds them (pads = finfo.min, last valid frame = finfo.max)."""
    keys = np.array(scores, dtype=np.float32)
    t = keys.shape[1]
    keys[~lmask(lengths, t)] = np.finfo(np.float32).min
    keys[np.arange(keys.shape[0]), np.asarray(lengths) - 1] = np.finfo(np.float32).max
    return keys


def assert_select_equal_up_to_ties(idx, ref_idx, nmask, keys, sg=None, ref_sg=None):
    """Equality with the reference wherever the reference's answer is DEFINED.

    The reference ranks frames with torch's non-stable CPU sort, whose order among equal keys is an
    implementation detail of the sort kernel (it changes with row length / ISA).  Both groups
    (kept = mask True, pad slots = mask False) must be ascending and must hold the same multiset of
    keys as the reference; when the keys of a row are pairwise distinct this IS exact equality,
    which is asserted on top.
    """
    for b in range(idx.shape[0]):
        kb = int(nmask[b].sum())
        for lo, hi in ((0, kb), (kb, idx.shape[1])):
            mine, ref = idx[b, lo:hi], ref_idx[b, lo:hi]
            assert np.all(np.diff(mine) > 0), "group must be strictly ascending"
            np.testing.assert_array_equal(np.sort(keys[b, mine]), np.sort(keys[b, ref]))
        sel = np.sort(keys[b, ref_idx[b]])
        rest = np.delete(keys[b], ref_idx[b])
        tied = (np.unique(sel).size != sel.size) or (rest.size and rest.max() >= sel.min())
        if not tied:
            np.testing.assert_array_equal(idx[b], ref_idx[b])
            if sg is not None:
                np.testing.assert_array_equal(sg[b], ref_sg[b])


@pytest.mark.parametrize("name", SELECT)
def test_select(name):
    g = load(name)
    t = g["scores"].shape[1]
    mask = lmask(g["lengths"], t)
    nmask, idx, sg = orc.topk_select(g["scores"], mask, float(g["rate"]))
    np.testing.assert_array_equal(nmask, g["nugget_mask"])
    keys = select_keys(g["scores"], g["lengths"])
    assert_select_equal_up_to_ties(idx, g["indices"], nmask, keys, sg, g["scores_gather"])
    np.testing.assert_array_equal(sg, np.take_along_axis(g["scores"], idx, axis=1))
    table, pos = orc.type_table(t, idx, nmask)
    ref_table, ref_pos = orc.type_table(t, g["indices"], g["nugget_mask"])
    np.testing.assert_array_equal(ref_table, g["type_table"])      # type_table restatement itself
    np.testing.assert_array_equal(ref_pos, g["positions"])
    if np.array_equal(idx, g["indices"]):
        np.testing.assert_array_equal(table, g["type_table"])
        np.testing.assert_array_equal(pos, g["positions"])


@pytest.mark.parametrize("name", ["selectn_a", "selectn_b", "selectn_over"])
def test_select_given_counts(name):
    # apply_nugget_compression(..., n_nugget=) as the word-delimiter inference calls pass it (w2v_asr.py:381,452): the
    # counts are used as given, without the clamp to the valid frames (selectn_over: 8 nuggets of 6 valid frames)
    g = load(name)
    t = g["scores"].shape[1]
    mask = lmask(g["lengths"], t)
    nmask, idx, sg = orc.topk_select(g["scores"], mask, n_nugget=g["n_nugget"])
    np.testing.assert_array_equal(nmask, g["nugget_mask"])
    assert_select_equal_up_to_ties(idx, g["indices"], nmask, select_keys(g["scores"], g["lengths"]), sg, g["scores_gather"])
    ref_table, ref_pos = orc.type_table(t, g["indices"], g["nugget_mask"])
    np.testing.assert_array_equal(ref_table, g["type_table"])
    np.testing.assert_array_equal(ref_pos, g["positions"])


@pytest.mark.parametrize("name", CIF)
def test_cif(name):
    g = load(name)
    t = g["scores"].shape[1]
    mask = lmask(g["lengths"], t)
    alphas, q = orc.quantity_loss(torch.from_numpy(g["scores"]), torch.from_numpy(mask),
                                  torch.from_numpy(g["target_lengths"]))
    np.testing.assert_array_equal(alphas.numpy(), g["alphas"])
    np.testing.assert_allclose(q.numpy(), g["quantity_loss"], rtol=1e-6)
    resized = orc.resize(alphas, torch.from_numpy(g["target_lengths"]))
    np.testing.assert_array_equal(resized.numpy(), g["resized"])
    nmask, idx, block = orc.fire_scan(g["resized"], mask)
    np.testing.assert_array_equal(idx, g["nugget_indices"])
    np.testing.assert_array_equal(nmask, g["nugget_mask"])
    np.testing.assert_array_equal(block, g["block_attn_mask"])


def test_scan_handmade():
    g = load("cif_scan_handmade")
    mask = lmask(g["lengths"], g["alphas"].shape[1])
    nmask, idx, block = orc.fire_scan(g["alphas"], mask)
    np.testing.assert_array_equal(idx, g["nugget_indices"])
    np.testing.assert_array_equal(nmask, g["nugget_mask"])
    np.testing.assert_array_equal(block, g["block_attn_mask"])


@pytest.mark.parametrize("name", XATTN)
def test_xattn(name):
    g = load(name)
    q, k, v, sc = (torch.from_numpy(g[n]).requires_grad_() for n in ("q", "k", "v", "scores"))
    kmask = torch.from_numpy(lmask(g["key_lengths"], k.shape[2]))
    # xattn_drop_*: attention-probability dropout (my_transformer.py:595) with the reference's mask given
    drop = dict(drop_mask=torch.from_numpy(g["keep"]), drop_p=float(g["drop_p"])) if "keep" in g else {}
    o, p = orc.xattn_core(q, k, v, kmask, sc, bool(g["ilk"]), **drop)
    (o * torch.from_numpy(g["gout"])).sum().backward()
    tol = dict(rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(o.detach().numpy(), g["o"], **tol)
    np.testing.assert_allclose(p.detach().numpy(), g["p"], **tol)
    np.testing.assert_allclose(q.grad.numpy(), g["dq"], **tol)
    np.testing.assert_allclose(k.grad.numpy(), g["dk"], **tol)
    np.testing.assert_allclose(v.grad.numpy(), g["dv"], **tol)
    np.testing.assert_allclose(sc.grad.numpy(), g["dscores"], **tol)


@pytest.mark.parametrize("name", ["headmask_a", "headmask_ilk"])
def test_xattn_head_mask(name):
    # my_transformer.py:598-599 with the [1, h, 1, 1] factors get_head_mask produces (reference run: gen_head_mask)
    g = load(name)
    q, k, v, sc, hm = (torch.from_numpy(g[n]).requires_grad_() for n in ("q", "k", "v", "scores", "head_mask"))
    kmask = torch.from_numpy(lmask(g["key_lengths"], k.shape[2]))
    o, p = orc.xattn_core(q, k, v, kmask, sc, bool(g["ilk"]), head_mask=hm)
    (o * torch.from_numpy(g["gout"])).sum().backward()
    tol = dict(rtol=1e-5, atol=1e-6)
    for got, want in ((o.detach(), "o"), (p.detach(), "p"), (q.grad, "dq"), (k.grad, "dk"), (v.grad, "dv"),
                      (sc.grad, "dscores"), (hm.grad, "dhead_mask")):
        np.testing.assert_allclose(got.numpy(), g[want], err_msg=want, **tol)


@pytest.mark.parametrize("name", ["scaling_noscale", "scaling_invlayer"])
def test_xattn_scaling_switches(name):
    # scale_attn_weights=False / scale_attn_by_inverse_layer_idx=True with layer_idx 2 (my_transformer.py:538-545)
    g = load(name)
    q, k, v, sc = (torch.from_numpy(g[n]).requires_grad_() for n in ("q", "k", "v", "scores"))
    kmask = torch.from_numpy(lmask(g["key_lengths"], k.shape[2]))
    o, p = orc.xattn_core(q, k, v, kmask, sc, bool(g["ilk"]), scale=bool(g["scale_attn_weights"]),
                          inverse_layer_idx=int(g["layer_idx"]) if bool(g["inverse_layer_idx"]) else None)
    (o * torch.from_numpy(g["gout"])).sum().backward()
    tol = dict(rtol=1e-5, atol=1e-6)
    for got, want in ((o.detach(), "o"), (p.detach(), "p"), (q.grad, "dq"), (k.grad, "dk"), (v.grad, "dv"),
                      (sc.grad, "dscores")):
        np.testing.assert_allclose(got.numpy(), g[want], err_msg=want, **tol)


def test_scorer():
    g = load("scorer_small")
    x, w1, b1, w2, b2 = (torch.from_numpy(g[n]).requires_grad_() for n in ("x", "w1", "b1", "w2", "b2"))
    s = orc.scorer(x, w1, b1, w2, b2)
    (s * torch.from_numpy(g["gscores"])).sum().backward()
    tol = dict(rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(s.detach().numpy(), g["scores"], **tol)
    for name, t in (("dx", x), ("dw1", w1), ("db1", b1), ("dw2", w2), ("db2", b2)):
        np.testing.assert_allclose(t.grad.numpy(), g[name], **tol)


def test_gather_rows_negative_index():
    src = torch.arange(2 * 5 * 3, dtype=torch.float32).view(2, 5, 3).requires_grad_()
    idx = torch.tensor([[0, 3, -1], [4, -1, -1]])
    mem = orc.gather_rows(src, idx)
    assert torch.equal(mem[0, 2], src[0, 4]) and torch.equal(mem[1, 1], src[1, 4])
    mem.sum().backward()
    assert src.grad[1, 4, 0] == 3 and src.grad[0, 1, 0] == 0


# ------------------------------------------------------------------ (6) end-to-end call-site fixtures
E2E = ["e2e_asr", "e2e_simul", "e2e_simul_ilk"]


def e2e_w1(g):
    """W1 of the e2e fixtures is regenerated from its numpy PCG64 seed (tests/golden/gen_golden.py: e2e_w1) and
    verified against the stored probe."""
    d = g["x"].shape[-1]
    rng = np.random.default_rng(int(g["w1_seed"]))
    w1 = ((rng.random((d, d), dtype=np.float32) * 2 - 1) / np.float32(d ** 0.5)).astype(np.float32)
    np.testing.assert_array_equal(w1[::97, ::89], g["w1_probe"])
    return w1


def e2e_select(g, scores):
    """The reference's nugget selection at its call site, restated with the oracle: returns
    (mask [B,K] bool, idx [B,K] int64 (pads as the reference leaves them), extra dscores term or None)."""
    B, T = scores.shape
    mask = lmask(g["lengths"], T)
    if "quantity_loss" not in g:
        nmask, idx, _ = orc.topk_select(scores, mask, float(g["rate"]))
        return mask, nmask, idx
    s = torch.from_numpy(scores)
    alphas, _ = orc.quantity_loss(s, torch.from_numpy(mask), torch.from_numpy(g["target_lengths"]))
    resized = orc.resize(alphas, torch.from_numpy(g["target_lengths"]))
    nmask, idx, _ = orc.fire_scan(resized.numpy(), mask, with_block_mask=False)
    return mask, nmask, idx


@pytest.mark.parametrize("name", E2E)
def test_e2e_call_sites(name):
    """Whole-model run of the reference (tiny decoder, fake encoders): the oracle,
    chained over the recorded call-site tensors, must reproduce the reference's selection, every cross-attention
    output and gradient, the total scorer-logit gradient and the nugget_model parameter gradients."""
    g = load(name)
    w1 = e2e_w1(g)
    x = torch.from_numpy(g["x"])
    w = [torch.from_numpy(a).clone().requires_grad_() for a in (w1, g["b1"], g["w2"], g["b2"])]
    scores = orc.scorer(x, *w)
    np.testing.assert_allclose(scores.detach().numpy(), g["scores"], rtol=1e-5, atol=1e-6)
    mask, nmask, idx = e2e_select(g, g["scores"])
    pos = np.where(nmask, idx, -1)
    np.testing.assert_array_equal(pos, g["nugget_position"])
    # scorer logits handed to the decoder (differentiable gather; -1 addresses the last frame in the simul path)
    gidx = torch.from_numpy(idx if "quantity_loss" not in g else pos)
    sg = orc.gather_rows(scores, gidx)
    total = 0.0
    for l in range(int(g["n_layers"])):
        q, k, v = (torch.from_numpy(g[f"{n}{l}"]).clone().requires_grad_() for n in "qkv")
        key_mask = torch.from_numpy(g[f"mask{l}"][:, 0, 0, :] == 0)          # inverted additive mask: 0 = attend
        np.testing.assert_array_equal(key_mask.numpy(), nmask)
        o, _ = orc.xattn_core(q, k, v, key_mask, sg, limit_cross_attn=bool(g["use_ilk"]))
        np.testing.assert_allclose(o.detach().numpy(), g[f"o{l}"], rtol=1e-4, atol=1e-5)
        total = total + (o * torch.from_numpy(g[f"do{l}"])).sum()
        gq, gk, gv = torch.autograd.grad((o * torch.from_numpy(g[f"do{l}"])).sum(), (q, k, v), retain_graph=True)
        for got, ref in ((gq, g[f"dq{l}"]), (gk, g[f"dk{l}"]), (gv, g[f"dv{l}"])):
            np.testing.assert_allclose(got.numpy(), ref, rtol=1e-3, atol=2e-7)
    if "quantity_loss" in g:
        _, ql = orc.quantity_loss(scores, torch.from_numpy(mask), torch.from_numpy(g["target_lengths"]))
        np.testing.assert_allclose(ql.item(), float(g["quantity_loss"]), rtol=1e-5)
        total = total + 0.01 * ql
    scores.retain_grad()
    total.backward()
    np.testing.assert_allclose(scores.grad.numpy(), g["dscores"], rtol=1e-3, atol=1e-8)
    np.testing.assert_allclose(w[0].grad.numpy()[:64], g["dw1_rows64"], rtol=1e-3, atol=1e-8)
    np.testing.assert_allclose(w[1].grad.numpy(), g["db1"], rtol=1e-3, atol=1e-8)
    np.testing.assert_allclose(w[2].grad.numpy(), g["dw2"], rtol=1e-3, atol=1e-8)
    # db2 = sum of dscores: analytically 0 in the ASR path (softmax is shift invariant), so bound it by the terms
    np.testing.assert_allclose(w[3].grad.numpy(), g["db2"], rtol=1e-3, atol=1e-5 * np.abs(g["dscores"]).sum())


# ------------------------------------------------------------------ 8f-2: CIF weighted aggregation
@pytest.mark.parametrize("name", ["cifagg_201", "cifagg_202"])
def test_cif_aggregate_golden(name):
    """Reference call site W2VSimulModel.lm_cif_training (w2v_simul.py:558-606): aggregated memory, its mask, the
    fired positions, and the gradients w.r.t. the semantic states and the (resized) alphas."""
    g = load(name)
    alphas = torch.from_numpy(g["alphas"]).clone().requires_grad_()
    s = torch.from_numpy(g["s"]).clone().requires_grad_()
    cif, mask, pos = orc.cif_aggregate(alphas, s)
    np.testing.assert_array_equal(pos.numpy(), g["positions"])
    np.testing.assert_array_equal(mask.numpy(), g["cif_mask"])
    np.testing.assert_allclose(cif.detach().numpy(), g["cif"], rtol=1e-6, atol=1e-7)
    (cif * torch.from_numpy(g["dcif"])).sum().backward()
    np.testing.assert_allclose(s.grad.numpy(), g["ds"], rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(alphas.grad.numpy(), g["dalphas"], rtol=1e-4, atol=1e-8)


# ------------------------------------------------------------------ 8f-3: type embedding added to the encoder states
@pytest.mark.parametrize("name", ["typeembed_a", "typeembed_b"])
def test_type_embed_golden(name):
    """oracle select -> type table -> x + E[table], and its gradients, against the reference call site
    (w2v_asr.py:555-560)."""
    g = load(name)
    x = torch.from_numpy(g["x"]).requires_grad_()
    emb = torch.from_numpy(g["emb"]).requires_grad_()
    T = x.shape[1]
    nmask, idx, _ = orc.topk_select(g["scores"], lmask(g["lengths"], T), rate=float(g["rate"]))
    table, _ = orc.type_table(T, idx, nmask)
    np.testing.assert_array_equal(table, g["table"])
    out = orc.type_embed_add(x, torch.from_numpy(table), emb)
    np.testing.assert_array_equal(out.detach().numpy(), g["out"])
    (out * torch.from_numpy(g["gout"])).sum().backward()
    np.testing.assert_array_equal(x.grad.numpy(), g["dx"])
    np.testing.assert_allclose(emb.grad.numpy(), g["demb"], rtol=1e-5, atol=1e-6)


# ------------------------------------------------------------------ round 2: resize gradient, valid indices, dropout mask
@pytest.mark.parametrize("name", ["resize_grad_plain", "resize_grad_fixup"])
def test_resize_gradient_golden(name):
    """resize differentiated as lm_cif_training does (w2v_simul.py:558): oracle autograd == reference autograd."""
    g = load(name)
    a = torch.from_numpy(g["alphas"]).clone().requires_grad_()
    r = orc.resize(a, torch.from_numpy(g["target_lengths"]), differentiable=True)
    np.testing.assert_array_equal(r.detach().numpy(), g["resized"])
    (r * torch.from_numpy(g["gout"])).sum().backward()
    np.testing.assert_allclose(a.grad.numpy(), g["dalphas"], rtol=1e-5, atol=1e-6)
    if name.endswith("fixup"):
        assert (g["resized"] != (g["alphas"] * (g["target_lengths"] / g["alphas"].sum(1))[:, None]).astype(np.float32)).any()


def test_valid_mask_indices_golden():
    g = load("valid_indices")
    for tag in "abcd":
        idx, kmax, n = orc.valid_mask_indices(g[f"mask_{tag}"], int(g[f"pad_{tag}"]))
        np.testing.assert_array_equal(idx, g[f"idx_{tag}"])
        assert kmax == int(g[f"kmax_{tag}"])
        np.testing.assert_array_equal(n, g[f"n_{tag}"])


def test_dropout_keep_mask_statistics():
    """The kernel's counter-based mask (restated from star_nugget.h): keep rate within 3 sigma of 1 - p_eff, streams
    of different (b, h), seeds and offsets differ, and the same arguments give the same mask."""
    p = 0.1
    m, p_eff = orc.dropout_keep_mask(1234, 7, 3, 4, 64, 200, p)
    assert abs(p_eff - p) < 1e-6
    n = m.size
    assert abs(m.mean() - (1 - p_eff)) < 3 * (p_eff * (1 - p_eff) / n) ** 0.5
    per = m.reshape(12, -1)
    sig = (p_eff * (1 - p_eff) / per.shape[1]) ** 0.5
    assert (np.abs(per.mean(1) - (1 - p_eff)) < 4.5 * sig).all()
    assert all((per[i] != per[j]).mean() > 0.1 for i in range(12) for j in range(i))
    m2, _ = orc.dropout_keep_mask(1234, 8, 3, 4, 64, 200, p)
    m3, _ = orc.dropout_keep_mask(1235, 7, 3, 4, 64, 200, p)
    assert (m != m2).mean() > 0.1 and (m != m3).mean() > 0.1
    m4, _ = orc.dropout_keep_mask(1234, 7, 3, 4, 64, 200, p)
    assert np.array_equal(m, m4)
    assert orc.dropout_keep_mask(1, 2, 1, 1, 4, 4, 0.0)[0].all()


# ------------------------------------------------------------------ 8f-3, second half: final LayerNorm + nugget row gather
@pytest.mark.parametrize("name", ["gatherln_a", "gatherln_b"])
def test_gather_layernorm_golden(name):
    """oracle ln_f + gather against the reference's MyGPT2Model.ln_f (my_transformer.py:945) and the row gather of
    w2v_asr.py:566-567, forward and all three gradients."""
    g = load(name)
    h = torch.from_numpy(g["hidden"]).requires_grad_()
    w = torch.from_numpy(g["weight"]).requires_grad_()
    b = torch.from_numpy(g["bias"]).requires_grad_()
    out = orc.gather_layernorm(h, torch.from_numpy(g["idx"]), w, b, float(g["eps"]))
    np.testing.assert_allclose(out.detach().numpy(), g["out"], rtol=1e-6, atol=1e-6)
    (out * torch.from_numpy(g["gout"])).sum().backward()
    np.testing.assert_allclose(h.grad.numpy(), g["dhidden"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(w.grad.numpy(), g["dweight"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(b.grad.numpy(), g["dbias"], rtol=1e-5, atol=1e-6)
