"""Generate golden vectors for the nugget hot path FROM THE TRUE REFERENCE.

Run ONLY in the build container (the reference does not travel to the GPU box):

    PYTHONPATH=/root/reference PYTHONDONTWRITEBYTECODE=1 HF_HUB_OFFLINE=1 \
        python tests/golden/gen_golden.py

It imports the reference's own functions (SURVEY.md section 8c recipe: the two absent
packages the reference imports but never uses on this path, torchaudio and
audiolm_pytorch, get empty in-memory placeholder modules), feeds them small seeded inputs
and stores inputs + reference outputs as .npz data files next to this script.  Nothing from
the reference's source text is stored -- only arrays.
"""
import os
import sys
import types
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("STAR_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
sys.dont_write_bytecode = True
os.environ.setdefault("HF_HUB_OFFLINE", "1")

import transformers  # noqa: E402,F401  (must be imported before the placeholders)

for name, attrs in (("torchaudio", {}), ("torchaudio.functional", {"resample": None}),
                    ("audiolm_pytorch", {}), ("audiolm_pytorch.version", {"__version__": "1.5.7"})):
    if name not in sys.modules:
        mod = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(mod, k, v)
        sys.modules[name] = mod

from src.models import w2v_asr, w2v_simul, my_transformer  # noqa: E402

torch.manual_seed(0)
torch.set_num_threads(1)


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, {k: v.shape for k, v in out.items()})


def len_mask(lengths, t):
    return torch.arange(t)[None, :] < torch.tensor(lengths)[:, None]


# ------------------------------------------------------------------ a2 + a3: top-k select
def gen_select():
    asr = w2v_asr.W2VASRModel
    cases = []
    g = torch.Generator().manual_seed(11)
    # (a) config-1 shaped
    cases.append(("c1", torch.randn(4, 250, generator=g), [250, 230, 200, 170], 4))
    # (b) heavy ties
    cases.append(("ties", (torch.randn(3, 37, generator=g) * 2).round() / 2, [37, 30, 11], 3))
    # (c) n_tok = 1 and K > n_tok rows (pad slots reach into padded frames)
    cases.append(("short", torch.randn(4, 40, generator=g), [1, 40, 7, 2], 2))
    # (d) rate 12, T not a multiple of 64
    cases.append(("r12", torch.randn(5, 101, generator=g), [101, 99, 64, 65, 13], 12))
    # (e) signed zeros and duplicates of the extreme values
    z = torch.zeros(2, 16)
    z[0, ::2] = -0.0
    z[0, 3] = 1.0
    z[1, 5:9] = torch.tensor([-0.0, 0.0, -0.0, 0.0])
    z[1, :5] = -1.0
    cases.append(("zeros", z, [16, 12], 4))
    # (f) rate 8, rate 2
    cases.append(("r8", torch.randn(3, 130, generator=g), [130, 128, 67], 8))
    cases.append(("r2", torch.randn(3, 64, generator=g), [64, 63, 33], 2))
    for name, scores, lengths, rate in cases:
        t = scores.shape[1]
        mask = len_mask(lengths, t)
        me = SimpleNamespace(nugget_compress_rate=rate)
        nmask, idx, sg = asr.apply_nugget_compression(me, torch.zeros(len(lengths), t, 2), mask,
                                                      nugget_scores=scores.clone())
        table, pos = asr.add_nugget_type_embedding(torch.zeros(len(lengths), t, 2), idx, nmask)
        save("select_" + name, scores=scores, lengths=np.array(lengths), rate=np.array(rate),
             nugget_mask=nmask, indices=idx, scores_gather=sg, type_table=table, positions=pos)
    # nugget-count sweep (w2v_asr.py:219-221)
    n = torch.arange(1, 3001)
    rows = {}
    for rate in (2, 3, 4, 8, 12, 18, 30):
        k = (n * (1 / rate) + 0.99).to(torch.int64)
        k[k > n] = n[k > n]
        rows[f"r{rate}"] = k
    save("select_counts", n=n, **rows)


# ------------------------------------------------------------------ a5 + a6 + a7: CIF path
def gen_cif():
    sim = w2v_simul.W2VSimulModel
    me = SimpleNamespace(get_valid_mask_indices=sim.get_valid_mask_indices)
    g = torch.Generator().manual_seed(23)
    cases = []
    cases.append(("plain", torch.randn(4, 48, generator=g), [48, 40, 31, 17], [6, 5, 4, 3]))
    # rows that need fix-up rounds: few valid frames, large targets, spiky scores
    s = torch.randn(4, 24, generator=g)
    s[0, 3] = 6.0
    s[1, :] = -4.0
    s[1, 5] = 5.0
    s[1, 9] = 4.0
    s[2, 2] = 8.0
    cases.append(("fixup", s, [24, 20, 6, 12], [9, 7, 5, 4]))
    cases.append(("ragged", torch.randn(6, 70, generator=g) - 1.0, [70, 69, 64, 33, 5, 1], [8, 7, 7, 4, 2, 1]))
    for name, scores, lengths, tgt in cases:
        t = scores.shape[1]
        mask = len_mask(lengths, t)
        tl = torch.tensor(tgt)
        alphas = torch.sigmoid(scores) * mask
        q = (((alphas / 1).sum(dim=1) - tl) ** 2).mean()
        resized = sim.resize(me, alphas.clone(), tl)
        nmask, idx, block = sim.apply_threshold_yield_nugget(me, resized.clone().detach(), mask)
        save("cif_" + name, scores=scores, lengths=np.array(lengths), target_lengths=tl,
             alphas=alphas, quantity_loss=q, resized=resized, nugget_mask=nmask,
             nugget_indices=idx, block_attn_mask=block)
    # scan-only cases on hand-made alphas (values > 1, carry past the last valid frame)
    a = torch.tensor([[0.5, 0.6, 2.5, 0.0, 0.1, 0.0, 0.0, 0.0],
                      [0.3, 0.3, 0.3, 0.3, 0.3, 0.3, 0.3, 0.3],
                      [1.0, 1.0, 0.999999, 1e-7, 0.5, 0.5, 0.0, 0.0]], dtype=torch.float32)
    lengths = [5, 8, 6]
    mask = len_mask(lengths, 8)
    nmask, idx, block = sim.apply_threshold_yield_nugget(me, a.clone(), mask)
    save("cif_scan_handmade", alphas=a, lengths=np.array(lengths), nugget_mask=nmask,
         nugget_indices=idx, block_attn_mask=block)


# ------------------------------------------------------------------ a8: score-biased attention core
def gen_xattn():
    att = my_transformer.MyGPT2Attention
    g = torch.Generator().manual_seed(37)
    cases = [("small", 2, 2, 5, 7, 8, [7, 4], False),
             ("ilk", 2, 2, 6, 7, 8, [7, 5], True),
             ("ilk_wide", 1, 3, 4, 12, 16, [9], True),
             ("hd96", 2, 2, 9, 21, 96, [21, 13], False)]
    for name, b, h, lq, lk, hd, klen, ilk in cases:
        q = torch.randn(b, h, lq, hd, generator=g).requires_grad_()
        k = torch.randn(b, h, lk, hd, generator=g).requires_grad_()
        v = torch.randn(b, h, lk, hd, generator=g).requires_grad_()
        sc = torch.randn(b, lk, generator=g).requires_grad_()
        gout = torch.randn(b, h, lq, hd, generator=g)
        kmask = len_mask(klen, lk)
        add = (1.0 - kmask[:, None, None, :].to(torch.float32)) * torch.finfo(torch.float32).min
        me = SimpleNamespace(scale_attn_weights=True, scale_attn_by_inverse_layer_idx=False,
                             is_cross_attention=True, layer_idx=0, attn_dropout=torch.nn.Identity())
        o, p = att._attn(me, q, k, v, add, None, sc, limit_cross_attn=ilk)
        (o * gout).sum().backward()
        save("xattn_" + name, q=q, k=k, v=v, scores=sc, key_lengths=np.array(klen), ilk=np.array(ilk),
             gout=gout, o=o, p=p, dq=q.grad, dk=k.grad, dv=v.grad, dscores=sc.grad)


# ------------------------------------------------------------------ a1: scorer (module as built at w2v_asr.py:113-117)
def gen_scorer():
    torch.manual_seed(5)
    d = 32
    from torch import nn
    model = nn.Sequential(nn.Linear(d, d), nn.ReLU(inplace=True), nn.Linear(d, 1))
    x = torch.randn(3, 19, d, requires_grad=True)
    s = model(x).squeeze(-1)
    gs = torch.randn(3, 19)
    (s * gs).sum().backward()
    save("scorer_small", x=x, w1=model[0].weight, b1=model[0].bias, w2=model[2].weight, b2=model[2].bias,
         scores=s, gscores=gs, dx=x.grad, dw1=model[0].weight.grad, db1=model[0].bias.grad,
         dw2=model[2].weight.grad, db2=model[2].bias.grad)


# ------------------------------------------------------------------ (6) end-to-end call-site fixtures
def _compat_patches():
    """transformers >= 5 dropped helpers that the 4.34 pin provided (API drift): restore their
    4.34 semantics on the reference classes (head split/merge = view + permute, no head mask)."""
    A = my_transformer.MyGPT2Attention
    if not hasattr(A, "_split_heads"):
        A._split_heads = lambda self, t, nh, hd: t.view(t.size()[:-1] + (nh, hd)).permute(0, 2, 1, 3)
    if not hasattr(A, "_merge_heads"):
        A._merge_heads = lambda self, t, nh, hd: t.permute(0, 2, 1, 3).contiguous().view(t.size()[:-3] + (t.size(-2), nh * hd))
    M = my_transformer.MyGPT2Model
    M.get_head_mask = lambda self, head_mask, n_layer, *a, **k: [None] * n_layer


class _FakeEncoder(torch.nn.Module):
    def __init__(self, x, mask):
        super().__init__()
        self.x, self.mask = x, mask

    def forward(self, audio_seq, attention_mask=None):
        return SimpleNamespace(last_hidden_state=self.x), self.mask


class _FakeSemantic(torch.nn.Module):
    def __init__(self, d, g):
        super().__init__()
        self.type_embed = torch.nn.Embedding(2, d)
        self.proj = torch.nn.Linear(d, d)
        with torch.no_grad():
            self.type_embed.weight.copy_(torch.randn(2, d, generator=g) * 0.1)
            self.proj.weight.copy_(torch.randn(d, d, generator=g) / d ** 0.5)
            self.proj.bias.zero_()

    def forward(self, inputs_embeds=None, attention_mask=None):
        return SimpleNamespace(last_hidden_state=torch.tanh(self.proj(inputs_embeds)))


def e2e_w1(seed, d):
    rng = np.random.default_rng(seed)
    return ((rng.random((d, d), dtype=np.float32) * 2 - 1) / np.float32(d ** 0.5)).astype(np.float32)


def _run_e2e(kind, use_ilk, seed):
    from transformers import GPT2Config
    _compat_patches()
    g = torch.Generator().manual_seed(seed)
    d, B, T, L, V, nl = 512, 2, 40, 12, 37, 2
    lengths = [40, 29]
    tlen = [12, 9]
    mask = len_mask(lengths, T)
    x = torch.randn(B, T, d, generator=g)
    x = (x - x.mean(-1, keepdim=True)) / x.std(-1, keepdim=True)
    cls = w2v_asr.W2VASRModel if kind == "asr" else w2v_simul.W2VSimulModel
    m = cls.__new__(cls)
    torch.nn.Module.__init__(m)
    m.disable_nuggets = False
    m.nugget_compress_rate = 4
    m.nugget_model = torch.nn.Sequential(torch.nn.Linear(d, d), torch.nn.ReLU(inplace=True), torch.nn.Linear(d, 1))
    with torch.no_grad():
        for prm in m.nugget_model.parameters():
            prm.copy_((torch.rand(prm.shape, generator=g) * 2 - 1) / d ** 0.5)
        # W1 (1 MB) is not stored: it is regenerated from a numpy PCG64 stream, with a probe to verify it
        m.nugget_model[0].weight.copy_(torch.from_numpy(e2e_w1(seed, d)))
        if kind == "simul":
            m.nugget_model[2].bias.fill_(-1.0)      # sigmoid(scores) ~ 0.27: sum of alphas in the range of L
    m.encoder = _FakeEncoder(x, mask)
    m.semantic_encoder = _FakeSemantic(d, g)
    cfg = GPT2Config(n_embd=d, vocab_size=V, num_labels=V, n_positions=64, pad_token_id=0, bos_token_id=1, eos_token_id=2,
                     n_layer=nl, n_head=8, add_cross_attention=True, attn_pdrop=0.0, resid_pdrop=0.0, embd_pdrop=0.0)
    torch.manual_seed(seed)
    if kind == "asr":
        m.lm_decoder = my_transformer.MyGPT2ForLM(cfg)
    else:
        m.limit_cross_attention = use_ilk
        m.lm_decoder = my_transformer.MyGPT2ForLM(cfg, limit_cross_attn=use_ilk)
    for mod in m.lm_decoder.modules():          # 4.34 registered the causal buffer on self-attention modules
        if isinstance(mod, my_transformer.MyGPT2Attention) and not getattr(mod, "is_cross_attention", False) \
                and not hasattr(mod, "bias"):
            mod.register_buffer("bias", torch.tril(torch.ones(64, 64, dtype=torch.bool)).view(1, 1, 64, 64), persistent=False)
    m.train()

    # record every cross-attention core call: inputs, output and (through hooks) their gradients
    rec = []
    orig_attn = my_transformer.MyGPT2Attention._attn

    def spy(self, query, key, value, attention_mask=None, head_mask=None, scorer_logits=None, limit_cross_attn=False, **kw):
        out = orig_attn(self, query, key, value, attention_mask, head_mask, scorer_logits=scorer_logits,
                        limit_cross_attn=limit_cross_attn, **kw)
        if getattr(self, "is_cross_attention", False):
            e = {"q": query, "k": key, "v": value, "o": out[0], "mask": attention_mask, "ilk": bool(limit_cross_attn)}
            for t in (query, key, value, out[0]):
                t.retain_grad()
            rec.append(e)
        return out

    my_transformer.MyGPT2Attention._attn = spy
    spy_sel = {}
    orig_sel = cls.apply_nugget_compression if kind == "asr" else None
    try:
        text = torch.randint(3, V, (B, L), generator=g)
        lm_mask = len_mask(tlen, L)
        labels = text.masked_fill(~lm_mask, -100)
        # keep handles on the intermediate tensors of the model method
        orig_nm_forward = m.nugget_model.forward
        keep = {}

        def nm_forward(inp):
            out = orig_nm_forward(inp)
            out.retain_grad()
            keep["scores3"] = out
            return out

        m.nugget_model.forward = nm_forward
        misc = {}
        loss, misc = m.lm_nugget_training(None, mask, text, lm_mask.long(), labels, misc)
        loss.backward()
    finally:
        my_transformer.MyGPT2Attention._attn = orig_attn
    nm = m.nugget_model
    scores = keep["scores3"].detach().squeeze(-1)
    out = dict(x=x, lengths=np.array(lengths), target_lengths=np.array(tlen), rate=4, use_ilk=int(use_ilk),
               w1_seed=seed, w1_probe=nm[0].weight.detach()[::97, ::89], b1=nm[0].bias, w2=nm[2].weight, b2=nm[2].bias,
               scores=scores, dscores=keep["scores3"].grad.squeeze(-1),
               dw1_rows64=nm[0].weight.grad[:64], db1=nm[0].bias.grad, dw2=nm[2].weight.grad, db2=nm[2].bias.grad,
               loss=loss.detach(), nugget_position=misc["nugget_indices"], n_layers=len(rec))
    if "quantity_loss" in misc:
        out["quantity_loss"] = np.float32(misc["quantity_loss"])
    for i, e in enumerate(rec):
        out[f"q{i}"], out[f"k{i}"], out[f"v{i}"], out[f"o{i}"] = e["q"], e["k"], e["v"], e["o"]
        out[f"dq{i}"], out[f"dk{i}"], out[f"dv{i}"], out[f"do{i}"] = e["q"].grad, e["k"].grad, e["v"].grad, e["o"].grad
        out[f"mask{i}"] = e["mask"]
    save(f"e2e_{kind}" + ("_ilk" if use_ilk else ""), **out)


def gen_cif_agg():
    """the reference's CIF weighted aggregation at its call site
    (W2VSimulModel.lm_cif_training, w2v_simul.py:536-621), run with a tiny real decoder: records the resized alphas,
    the semantic states, the aggregated memory handed to the decoder, and the gradients of all three."""
    from transformers import GPT2Config
    _compat_patches()
    for seed, lengths, tlen in ((201, [40, 29], [12, 9]), (202, [37, 40, 5], [10, 14, 2])):
        g = torch.Generator().manual_seed(seed)
        d, T, L, V, nl = 512, 40, 14, 37, 1
        B = len(lengths)
        mask = len_mask(lengths, T)
        x = torch.randn(B, T, d, generator=g)
        cls = w2v_simul.W2VSimulModel
        m = cls.__new__(cls)
        torch.nn.Module.__init__(m)
        m.nugget_model = torch.nn.Sequential(torch.nn.Linear(d, d), torch.nn.ReLU(inplace=True), torch.nn.Linear(d, 1))
        with torch.no_grad():
            for prm in m.nugget_model.parameters():
                prm.copy_((torch.rand(prm.shape, generator=g) * 2 - 1) / d ** 0.5)
        m.encoder = _FakeEncoder(x, mask)
        sem = _FakeSemantic(d, g)
        keep = {}
        orig_sem = sem.forward

        def sem_forward(inputs_embeds=None, attention_mask=None):
            out = orig_sem(inputs_embeds=inputs_embeds, attention_mask=attention_mask)
            out.last_hidden_state.retain_grad()
            keep["s"] = out.last_hidden_state
            return out

        sem.forward = sem_forward
        m.semantic_encoder = sem
        cfg = GPT2Config(n_embd=d, vocab_size=V, num_labels=V, n_positions=64, pad_token_id=0, bos_token_id=1,
                         eos_token_id=2, n_layer=nl, n_head=8, add_cross_attention=True, attn_pdrop=0.0, resid_pdrop=0.0,
                         embd_pdrop=0.0)
        torch.manual_seed(seed)
        m.lm_decoder = my_transformer.MyGPT2ForLM(cfg)
        for mod in m.lm_decoder.modules():
            if isinstance(mod, my_transformer.MyGPT2Attention) and not getattr(mod, "is_cross_attention", False) \
                    and not hasattr(mod, "bias"):
                mod.register_buffer("bias", torch.tril(torch.ones(64, 64, dtype=torch.bool)).view(1, 1, 64, 64),
                                    persistent=False)
        m.train()
        orig_resize = m.resize

        def resize_spy(alphas, target_lengths, noise=0.0, threshold=1.0):
            out = orig_resize(alphas, target_lengths=target_lengths, noise=noise, threshold=threshold)
            out.retain_grad()
            keep["alphas"] = out
            return out

        m.resize = resize_spy
        orig_dec = m.lm_decoder.forward

        def dec_spy(*a, **kw):
            kw["encoder_hidden_states"].retain_grad()
            keep["cif"] = kw["encoder_hidden_states"]
            keep["cif_mask"] = kw["encoder_attention_mask"]
            return orig_dec(*a, **kw)

        m.lm_decoder.forward = dec_spy
        text = torch.randint(3, V, (B, L), generator=g)
        lm_mask = len_mask(tlen, L)
        labels = text.masked_fill(~lm_mask, -100)
        misc = {}
        loss, misc = m.lm_cif_training(torch.zeros(B, 1), mask, text, lm_mask.long(), labels, misc)
        loss.backward()
        save(f"cifagg_{seed}", alphas=keep["alphas"], s=keep["s"], lengths=np.array(lengths),
             cif=keep["cif"], cif_mask=keep["cif_mask"], positions=misc["nugget_indices"],
             dcif=keep["cif"].grad, ds=keep["s"].grad, dalphas=keep["alphas"].grad, loss=loss.detach())


def gen_e2e():
    _run_e2e("asr", False, 101)
    _run_e2e("simul", False, 102)
    _run_e2e("simul", True, 103)


# ------------------------------------------------------------------ 8f-3: type embedding added to the encoder states
def gen_type_embed():
    """The reference lines `table, pos = self.add_nugget_type_embedding(x, idx, mask)` and
    `inputs_embed = x + self.semantic_encoder.type_embed(table)` (w2v_asr.py:559-560), forward and backward, with the
    reference's own compression picking the nuggets; type_embed = nn.Embedding(2, d) as in my_transformer.py:761."""
    asr = w2v_asr.W2VASRModel
    for name, (B, T, d, lengths, rate, seed) in {"a": (3, 50, 64, [50, 41, 7], 4, 31),
                                                 "b": (2, 33, 128, [33, 20], 12, 32)}.items():
        g = torch.Generator().manual_seed(seed)
        x = torch.randn(B, T, d, generator=g, requires_grad=True)
        scores = torch.randn(B, T, generator=g)
        mask = len_mask(lengths, T)
        emb = torch.nn.Embedding(2, d)
        with torch.no_grad():
            emb.weight.copy_(torch.randn(2, d, generator=g) * 0.1)
        ns = SimpleNamespace(nugget_compress_rate=rate, nugget_model=None)
        nmask, idx, _ = asr.apply_nugget_compression(ns, x, mask, nugget_scores=scores)
        table, pos = asr.add_nugget_type_embedding(x, idx, nmask)
        out = x + emb(table)
        gout = torch.randn(B, T, d, generator=g)
        (out * gout).sum().backward()
        save(f"typeembed_{name}", x=x, lengths=np.array(lengths), scores=scores, rate=rate, emb=emb.weight, table=table,
             out=out, gout=gout, dx=x.grad, demb=emb.weight.grad)


# ------------------------------------------------------------------ round 2 additions
def gen_resize_grad():
    """resize (w2v_simul.py:504-535) WITH its autograd gradient, as lm_cif_training differentiates it (w2v_simul.py:558):
    alphas = sigmoid(scores) * mask with requires_grad, a random cotangent on the resized alphas."""
    sim = w2v_simul.W2VSimulModel
    me = SimpleNamespace()
    g = torch.Generator().manual_seed(61)
    cases = [("plain", torch.randn(4, 48, generator=g), [48, 40, 31, 17], [6, 5, 4, 3])]
    s = torch.randn(5, 24, generator=g)
    s[0, 3] = 6.0
    s[1, :] = -4.0
    s[1, 5] = 5.0
    s[1, 9] = 4.0
    s[2, 2] = 8.0
    s[4, 7] = 9.0
    s[4, 11] = 7.5
    cases.append(("fixup", s, [24, 20, 6, 12, 16], [9, 7, 5, 4, 11]))
    for name, scores, lengths, tgt in cases:
        t = scores.shape[1]
        mask = len_mask(lengths, t)
        tl = torch.tensor(tgt)
        alphas = (torch.sigmoid(scores) * mask).detach().requires_grad_()
        resized = sim.resize(me, alphas, tl)
        gout = torch.randn(resized.shape, generator=g)
        (resized * gout).sum().backward()
        save("resize_grad_" + name, alphas=alphas, lengths=np.array(lengths), target_lengths=tl, resized=resized,
             gout=gout, dalphas=alphas.grad)


def gen_valid_indices():
    """get_valid_mask_indices (w2v_simul.py:196-220) on masks whose last element is False / True, 1-D and 2-D, with
    the default and a non-default padding value (a forced last-frame fire must not leak in)."""
    sim = w2v_simul.W2VSimulModel
    g = torch.Generator().manual_seed(67)
    m2 = torch.rand(5, 23, generator=g) < 0.3
    m2[0, -1] = False
    m2[1, -1] = True
    m2[2, :] = False
    m2[2, 4] = True
    m2[3, -3:] = False
    m1 = torch.rand(17, generator=g) < 0.4
    m1[-1] = False
    out = {}
    for tag, m, pv in (("a", m2, -1), ("b", m2, -7), ("c", m1, -1), ("d", m1, 99)):
        idx, kmax, n = sim.get_valid_mask_indices(m, padding_value=pv)
        out[f"mask_{tag}"], out[f"pad_{tag}"], out[f"idx_{tag}"] = m, np.int64(pv), idx
        out[f"kmax_{tag}"], out[f"n_{tag}"] = np.int64(kmax), n
    save("valid_indices", **out)


def gen_xattn_drop():
    """MyGPT2Attention._attn (my_transformer.py:535-603) with attention-probability dropout (:595) under a GIVEN keep
    mask: `attn_dropout` is replaced by a module that multiplies by mask / (1 - p), which is what nn.Dropout(p) does
    with the mask it draws.  Pins where the mask enters the forward and the gradients."""
    att = my_transformer.MyGPT2Attention
    g = torch.Generator().manual_seed(71)

    class GivenDropout(torch.nn.Module):
        def __init__(self, mask, p):
            super().__init__()
            self.mask, self.p = mask, p

        def forward(self, w):
            return w * self.mask / (1.0 - self.p)

    for name, b, h, lq, lk, hd, klen, ilk, p in (("drop_a", 2, 2, 6, 9, 8, [9, 5], False, 0.25),
                                                 ("drop_ilk", 2, 3, 7, 10, 16, [10, 8], True, 0.1)):
        q = torch.randn(b, h, lq, hd, generator=g).requires_grad_()
        k = torch.randn(b, h, lk, hd, generator=g).requires_grad_()
        v = torch.randn(b, h, lk, hd, generator=g).requires_grad_()
        sc = torch.randn(b, lk, generator=g).requires_grad_()
        gout = torch.randn(b, h, lq, hd, generator=g)
        keep = (torch.rand(b, h, lq, lk, generator=g) >= p).to(torch.float32)
        kmask = len_mask(klen, lk)
        add = (1.0 - kmask[:, None, None, :].to(torch.float32)) * torch.finfo(torch.float32).min
        me = SimpleNamespace(scale_attn_weights=True, scale_attn_by_inverse_layer_idx=False,
                             is_cross_attention=True, layer_idx=0, attn_dropout=GivenDropout(keep, p))
        o, pw = att._attn(me, q, k, v, add, None, sc, limit_cross_attn=ilk)
        (o * gout).sum().backward()
        save("xattn_" + name, q=q, k=k, v=v, scores=sc, key_lengths=np.array(klen), ilk=np.array(ilk),
             keep=keep, drop_p=np.float32(p), gout=gout, o=o, p=pw, dq=q.grad, dk=k.grad, dv=v.grad, dscores=sc.grad)


def gen_gather_ln():
    """The tail of the reference's semantic encoder and the gather that follows it: `hidden_states = self.ln_f(
    hidden_states)` of MyGPT2Model.forward (my_transformer.py:945, ln_f = nn.LayerNorm(n_embd, eps=layer_norm_epsilon),
    :765) on a real MyGPT2Model instance, then `semantic_repr[torch.arange(B).unsqueeze(1), nugget_indices]`
    (w2v_asr.py:566-567) with the reference's own compression picking the rows (pad slots included); forward and the
    gradients into the hidden states and the LayerNorm parameters."""
    from transformers import GPT2Config
    asr = w2v_asr.W2VASRModel
    for name, (B, T, d, lengths, rate, seed) in {"a": (3, 50, 64, [50, 41, 7], 4, 41),
                                                 "b": (2, 37, 256, [37, 20], 2, 42)}.items():
        g = torch.Generator().manual_seed(seed)
        cfg = GPT2Config(n_embd=d, vocab_size=4, n_positions=64, n_layer=1, n_head=4, add_cross_attention=False)
        model = my_transformer.MyGPT2Model(cfg)
        ln_f = model.ln_f
        with torch.no_grad():
            ln_f.weight.copy_(1.0 + 0.3 * torch.randn(d, generator=g))
            ln_f.bias.copy_(0.2 * torch.randn(d, generator=g))
        hidden = (torch.randn(B, T, d, generator=g) * 1.7 + 0.4).requires_grad_()
        scores = torch.randn(B, T, generator=g)
        mask = len_mask(lengths, T)
        ns = SimpleNamespace(nugget_compress_rate=rate, nugget_model=None)
        nmask, idx, _ = asr.apply_nugget_compression(ns, hidden, mask, nugget_scores=scores)
        y = ln_f(hidden)
        out = y[torch.arange(B).unsqueeze(1), idx]
        gout = torch.randn(out.shape, generator=g)
        (out * gout).sum().backward()
        save(f"gatherln_{name}", hidden=hidden, idx=idx, weight=ln_f.weight, bias=ln_f.bias, eps=np.float64(ln_f.eps),
             out=out, gout=gout, dhidden=hidden.grad, dweight=ln_f.weight.grad, dbias=ln_f.bias.grad)


def gen_xattn_scaling():
    """MyGPT2Attention._attn with the two scaling switches off their defaults (my_transformer.py:538-545):
    scale_attn_weights=False (no 1/sqrt(hd)) and scale_attn_by_inverse_layer_idx=True with layer_idx = 2 (a further
    1/3), with the score bias and, in one case, the infinite-lookback mask."""
    att = my_transformer.MyGPT2Attention
    g = torch.Generator().manual_seed(113)
    for name, b, h, lq, lk, hd, klen, ilk, saw, inv, li in (("noscale", 2, 2, 5, 8, 8, [8, 5], False, False, False, 0),
                                                            ("invlayer", 2, 3, 6, 9, 16, [9, 7], True, True, True, 2)):
        q = (torch.randn(b, h, lq, hd, generator=g) * (0.4 if not saw else 1.0)).requires_grad_()
        k = torch.randn(b, h, lk, hd, generator=g).requires_grad_()
        v = torch.randn(b, h, lk, hd, generator=g).requires_grad_()
        sc = torch.randn(b, lk, generator=g).requires_grad_()
        gout = torch.randn(b, h, lq, hd, generator=g)
        kmask = len_mask(klen, lk)
        add = (1.0 - kmask[:, None, None, :].to(torch.float32)) * torch.finfo(torch.float32).min
        me = SimpleNamespace(scale_attn_weights=saw, scale_attn_by_inverse_layer_idx=inv,
                             is_cross_attention=True, layer_idx=li, attn_dropout=torch.nn.Identity())
        o, pw = att._attn(me, q, k, v, add, None, sc, limit_cross_attn=ilk)
        (o * gout).sum().backward()
        save("scaling_" + name, q=q, k=k, v=v, scores=sc, key_lengths=np.array(klen), ilk=np.array(ilk),
             scale_attn_weights=np.array(saw), inverse_layer_idx=np.array(inv), layer_idx=np.array(li),
             gout=gout, o=o, p=pw, dq=q.grad, dk=k.grad, dv=v.grad, dscores=sc.grad)


def gen_select_given_counts():
    """apply_nugget_compression with the nugget counts GIVEN (n_nugget=, as the word-delimiter-driven inference calls
    at w2v_asr.py:381 and :452 pass them: max(#delimiters, n_tok // 20 [+ 1])) instead of derived from the rate:
    no clamp to n_tok is applied on this path (w2v_asr.py:218-221 is skipped), so a count may exceed the valid frames
    and its kept set then reaches into padded frames."""
    asr = w2v_asr.W2VASRModel
    g = torch.Generator().manual_seed(97)
    for name, t, lengths, counts in (("a", 60, [60, 41, 17], [7, 3, 1]),
                                     ("b", 130, [130, 129, 64, 5], [9, 40, 4, 2]),
                                     ("over", 24, [24, 6, 10], [5, 8, 10])):          # 8 > 6 valid frames
        scores = torch.randn(len(lengths), t, generator=g)
        mask = len_mask(lengths, t)
        me = SimpleNamespace(nugget_compress_rate=4)
        nmask, idx, sg = asr.apply_nugget_compression(me, torch.zeros(len(lengths), t, 2), mask,
                                                      nugget_scores=scores.clone(), n_nugget=torch.tensor(counts))
        table, pos = asr.add_nugget_type_embedding(torch.zeros(len(lengths), t, 2), idx, nmask)
        save("selectn_" + name, scores=scores, lengths=np.array(lengths), n_nugget=np.array(counts),
             nugget_mask=nmask, indices=idx, scores_gather=sg, type_table=table, positions=pos)


def gen_head_mask():
    """MyGPT2Attention._attn (my_transformer.py:535-603) with a head mask: `attn_weights = attn_weights * head_mask`
    after the dropout (:598-599), in the shape HF's get_head_mask hands to a block ([1, h, 1, 1], one factor per head;
    the reference's own trainers pass None).  Pins that the factor enters after the softmax -- the returned weights,
    the output and every gradient carry it."""
    att = my_transformer.MyGPT2Attention
    g = torch.Generator().manual_seed(83)
    for name, b, h, lq, lk, hd, klen, ilk, hm in (("a", 2, 3, 5, 9, 8, [9, 6], False, [1.0, 0.0, 0.5]),
                                                  ("ilk", 2, 4, 7, 8, 16, [8, 5], True, [0.25, 1.0, 1.0, 0.0])):
        q = torch.randn(b, h, lq, hd, generator=g).requires_grad_()
        k = torch.randn(b, h, lk, hd, generator=g).requires_grad_()
        v = torch.randn(b, h, lk, hd, generator=g).requires_grad_()
        sc = torch.randn(b, lk, generator=g).requires_grad_()
        gout = torch.randn(b, h, lq, hd, generator=g)
        head_mask = torch.tensor(hm, dtype=torch.float32).view(1, h, 1, 1).requires_grad_()
        kmask = len_mask(klen, lk)
        add = (1.0 - kmask[:, None, None, :].to(torch.float32)) * torch.finfo(torch.float32).min
        me = SimpleNamespace(scale_attn_weights=True, scale_attn_by_inverse_layer_idx=False,
                             is_cross_attention=True, layer_idx=0, attn_dropout=torch.nn.Identity())
        o, pw = att._attn(me, q, k, v, add, head_mask, sc, limit_cross_attn=ilk)
        (o * gout).sum().backward()
        save("headmask_" + name, q=q, k=k, v=v, scores=sc, key_lengths=np.array(klen), ilk=np.array(ilk),
             head_mask=head_mask, gout=gout, o=o, p=pw, dq=q.grad, dk=k.grad, dv=v.grad, dscores=sc.grad,
             dhead_mask=head_mask.grad)


if __name__ == "__main__":
    only = sys.argv[1:]
    for fn in (gen_select, gen_cif, gen_xattn, gen_scorer, gen_e2e, gen_cif_agg, gen_type_embed, gen_resize_grad,
               gen_valid_indices, gen_xattn_drop, gen_gather_ln, gen_head_mask, gen_select_given_counts,
               gen_xattn_scaling):
        if not only or fn.__name__ in only:
            fn()
