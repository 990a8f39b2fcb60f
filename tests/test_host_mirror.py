"""Host-side mirror of the reference's operator interface (no GPU): module names, parameter / state_dict keys, and
the error behaviour of the drop-in functions."""
import pytest
import torch

import star_amd
from star_amd.nugget import NuggetCrossAttention, NuggetOpsMixin, NuggetScorer, score_biased_attn


def test_scorer_state_dict_matches_the_reference_sequential():
    """nugget_model = nn.Sequential(nn.Linear(h,h), nn.ReLU(inplace=True), nn.Linear(h,1)) (w2v_asr.py:113-117): the
    same keys and shapes, so checkpoints and the trainers' "nugget" optimizer grouping keep working."""
    d = 32
    ref = torch.nn.Sequential(torch.nn.Linear(d, d), torch.nn.ReLU(inplace=True), torch.nn.Linear(d, 1))
    mine = NuggetScorer(d)
    assert list(mine.state_dict().keys()) == list(ref.state_dict().keys()) == ["0.weight", "0.bias", "2.weight", "2.bias"]
    mine.load_state_dict(ref.state_dict(), strict=True)
    for k, v in ref.state_dict().items():
        assert torch.equal(mine.state_dict()[k], v)

    class Model(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.nugget_model = NuggetScorer(d)

    names = [n for n, _ in Model().named_parameters()]
    assert names == ["nugget_model.0.weight", "nugget_model.0.bias", "nugget_model.2.weight", "nugget_model.2.bias"]
    assert all("nugget" in n for n in names)                       # wav2vec_trainer.py:385-387 groups by substring


def test_cross_attention_parameter_names_and_conv1d_layout():
    """GPT2Attention(is_cross_attention=True): q_attn, c_attn, c_proj as HF Conv1D (weight [in, out])."""
    d = 48
    m = NuggetCrossAttention(d, 4)
    shapes = {n: tuple(p.shape) for n, p in m.named_parameters()}
    assert shapes == {"c_attn.weight": (d, 2 * d), "c_attn.bias": (2 * d,), "q_attn.weight": (d, d),
                      "q_attn.bias": (d,), "c_proj.weight": (d, d), "c_proj.bias": (d,)}
    with pytest.raises(ValueError):
        NuggetCrossAttention(50, 4)                                # embed_dim must be divisible by num_heads
    with pytest.raises(ValueError):
        m(torch.zeros(1, 2, d))                                    # cross-attention needs encoder_hidden_states
    # the reference's default decoder config (attn_pdrop 0.1, w2v_asr.py:134-145) is accepted: dropout runs inside the
    # kernels in training mode, so on CPU tensors the call reaches the "no CPU fallback" error, not a refusal
    drop = NuggetCrossAttention(d, 4, attn_pdrop=0.1)
    assert drop.attn_pdrop == pytest.approx(0.1) and drop.training
    with pytest.raises(star_amd.ops.StarDeviceError):
        drop(torch.zeros(1, 2, d), encoder_hidden_states=torch.zeros(1, 3, d))
    with pytest.raises(ValueError):
        NuggetCrossAttention(d, 4, attn_pdrop=1.0)


def test_active_rows_hint_is_single_use():
    """the active-row hint must not leak into a later forward/backward of the same scorer."""
    h = star_amd.ops.ActiveRows()
    rows = torch.arange(4, dtype=torch.int32)
    f1 = h._begin_forward()
    h.set(rows)                              # after the forward (what nugget_step_asr does): bound to f1
    f2 = h._begin_forward()                  # e.g. a second forward for a dense regulariser
    assert h._take(f2) is None               # ... runs dense
    assert h._take(f1) is rows               # the owner consumes it
    assert h._take(f1) is None and h.rows is None
    h.set(rows)                              # a second backward through the same forward (retain_graph): set again
    assert h._take(f1) is None and h._take(f2) is rows
    f3 = h._begin_forward()                  # the next step forgets to set it: dense
    assert h._take(f3) is None
    h.set(rows)
    h.set(None)
    assert h._take(f3) is None
    fresh = star_amd.ops.ActiveRows()
    fresh.set(rows)                          # before any forward: bound to the next one
    assert fresh._take(fresh._begin_forward()) is rows


def test_error_behaviour_without_a_gpu():
    q = torch.zeros(1, 2, 3, 64)
    k = torch.zeros(1, 2, 5, 64)
    with pytest.raises(NotImplementedError):
        score_biased_attn(q, k, k, head_mask=torch.ones(2))
    with pytest.raises(ValueError):
        score_biased_attn(q, k, k, scorer_logits=torch.zeros(1, 6))      # more logits than keys
    with pytest.raises(ValueError):
        score_biased_attn(q, k, k, attention_mask=torch.zeros(1, 1, 1, 4))
    with pytest.raises(star_amd.ops.StarDeviceError):
        score_biased_attn(q, k, k)                                      # CPU tensors: there is no fallback path
    with pytest.raises(NotImplementedError):
        NuggetOpsMixin().resize(torch.zeros(1, 4), torch.ones(1), noise=0.1)
    with pytest.raises(star_amd.ops.StarDeviceError):
        star_amd.share_scorer_logits(torch.zeros(2, 5), 3, 4)
    with pytest.raises(star_amd.ops.StarDeviceError):
        star_amd.ops.type_embed_add(torch.zeros(2, 3, 8), torch.zeros(2, 3, dtype=torch.long), torch.zeros(2, 8))


def _torch_attn(q, k, v):
    p = torch.softmax(q @ k.transpose(-1, -2) / q.shape[-1] ** 0.5, dim=-1)
    return p @ v


def test_share_encoder_kv_equals_per_layer_c_attn_cpu():
    """one projection for all layers == c_attn + split_heads of every layer
    (my_transformer.py:500-508), values and gradients (CPU: the gradients arrive through the copy fallback)."""
    torch.manual_seed(3)
    B, K, L, d, h, nl = 2, 7, 5, 32, 4, 3
    mods = [NuggetCrossAttention(d, h, layer_idx=i) for i in range(nl)]
    for m in mods:
        torch.nn.init.normal_(m.c_attn.bias, std=0.1)
    mem = torch.randn(B, K, d, requires_grad=True)
    qs = [torch.randn(B, h, L, d // h) for _ in range(nl)]
    gs = [torch.randn(B, h, L, d // h) for _ in range(nl)]

    def run(shared):
        for m in mods:
            m.zero_grad()
        mem.grad = None
        if shared:
            kvs = star_amd.share_encoder_kv(mods, mem)
            pairs = [(e.key, e.value) for e in kvs]
            assert all(e.grad_slot.layer == i for i, e in enumerate(kvs))
        else:
            pairs = []
            for m in mods:
                k, v = m.c_attn(mem).split(d, dim=2)
                pairs.append((m._split_heads(k), m._split_heads(v)))
        loss = sum((_torch_attn(q, k, v) * g).sum() for q, (k, v), g in zip(qs, pairs, gs))
        loss.backward()
        return ([t.detach().clone() for kv in pairs for t in kv], mem.grad.clone(),
                [m.c_attn.weight.grad.clone() for m in mods], [m.c_attn.bias.grad.clone() for m in mods])

    ref, got = run(False), run(True)
    for a, b in zip(ref[0], got[0]):
        assert a.shape == b.shape == (B, h, K, d // h)
        torch.testing.assert_close(b, a, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(got[1], ref[1], rtol=1e-4, atol=1e-5)
    for a, b in zip(ref[2] + ref[3], got[2] + got[3]):
        torch.testing.assert_close(b, a, rtol=1e-4, atol=1e-5)


def test_share_encoder_kv_unused_layer_and_errors():
    torch.manual_seed(4)
    B, K, d, h = 2, 5, 16, 2
    mods = [NuggetCrossAttention(d, h) for _ in range(2)]
    mem = torch.randn(B, K, d, requires_grad=True)
    kvs = star_amd.share_encoder_kv(mods, mem)
    (kvs[1].key.sum() * 2.0).backward()                    # layer 0 and V of layer 1 get no gradient: zeros, not garbage
    assert torch.count_nonzero(mods[0].c_attn.weight.grad) == 0
    assert torch.count_nonzero(mods[1].c_attn.weight.grad[:, d:]) == 0
    torch.testing.assert_close(mods[1].c_attn.bias.grad[:d], torch.full((d,), 2.0 * B * K))
    assert mem.grad is not None and torch.isfinite(mem.grad).all()
    with pytest.raises(ValueError):
        star_amd.share_encoder_kv([], mem)
    with pytest.raises(ValueError):
        star_amd.share_encoder_kv([NuggetCrossAttention(d, 2), NuggetCrossAttention(d, 4)], mem)
    with pytest.raises(ValueError):
        star_amd.ops.share_kv(mem, torch.zeros(d, 3 * d), torch.zeros(3 * d), 2, h)
    with pytest.raises(TypeError):
        star_amd.ops.share_kv(mem, torch.zeros(d, 4 * d, dtype=torch.bfloat16), torch.zeros(4 * d), 2, h)
