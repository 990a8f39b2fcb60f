"""star_scorer_fwd_static: the entry point that allocates nothing (its partial-score scratch is a file-scope
__device__ array inside the library) against star_scorer_fwd with a caller workspace -- same kernels, so the scores
must be bit-identical -- on two calls in a row, the second with a smaller input (the array keeps the first call's
contents).  Called through the C ABI."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _scores(L, fn_static, x, w1b, b1, w2, b2):
    from star_amd import _lib
    N, d = x.shape
    out = torch.full((N,), float("nan"), dtype=torch.float32, device=DEV)
    if fn_static:
        rc = L.star_scorer_fwd_static(x.data_ptr(), w1b.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), N, d,
                                      out.data_ptr(), None, _lib.stream_ptr())
    else:
        wsb = L.star_scorer_fwd_workspace(N, d)
        ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
        rc = L.star_scorer_fwd(x.data_ptr(), w1b.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), N, d,
                               out.data_ptr(), None, ws.data_ptr(), wsb, _lib.stream_ptr())
    _lib.check(rc, "star_scorer_fwd_static" if fn_static else "star_scorer_fwd")
    torch.cuda.synchronize()
    return out


def test_scorer_fwd_static_matches_workspace_variant():
    from star_amd import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(11)
    d = 512
    w1b = (torch.randn(d, d, generator=g) / d ** 0.5).to(torch.bfloat16).to(DEV)
    b1 = (torch.randn(d, generator=g) * 0.1).to(DEV)
    w2 = (torch.randn(d, generator=g) / d ** 0.5).to(DEV)
    b2 = torch.randn(1, generator=g).to(DEV)
    # first a large input, then a smaller one: stale partials of the first call must not leak into the second
    for N in (5000, 700, 1):
        x = torch.randn(N, d, generator=g).to(torch.bfloat16).to(DEV)
        ref = _scores(L, False, x, w1b, b1, w2, b2)
        got = _scores(L, True, x, w1b, b1, w2, b2)
        assert torch.isfinite(got).all()
        assert torch.equal(got, ref), f"N={N}: static-scratch scores differ from the workspace variant"
    # beyond the array: a clear error, no launch
    big = L.star_scorer_fwd_workspace(600_000, 1024)
    assert big > (4 << 20)
    x = torch.zeros(8, 1024, dtype=torch.bfloat16, device=DEV)
    rc = L.star_scorer_fwd_static(x.data_ptr(), x.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), 600_000,
                                  1024, x.data_ptr(), None, _lib.stream_ptr())
    assert rc == -3 and b"static array holds" in L.star_last_error()
