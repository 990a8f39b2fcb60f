"""CPU-side checks of the drop-in boundary: the shared library loads and exports every symbol that
include/*.h declare (star_nugget.h = the product ABI, star_harness.h = the harness-only fan-out stand-in), and the
ctypes table mirrors the headers (no compute calls: no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADERS = [os.path.join(ROOT, "include", n) for n in ("star_nugget.h", "star_harness.h")]


def declared_functions(headers=HEADERS):
    names = set()
    for h in headers:
        text = open(h).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(star_[a-z0-9_]+)\s*\(", text))
    return sorted(names)


def test_harness_stand_in_is_not_in_the_product_header():
    product = declared_functions(HEADERS[:1])
    assert not [n for n in product if "fanout" in n]
    assert declared_functions(HEADERS[1:]) == ["star_diag_fanout_bwd", "star_diag_fanout_fwd"]


def test_header_declares_the_expected_entry_points():
    names = declared_functions()
    for must in ("star_topk_select", "star_gather_rows_fwd", "star_gather_rows_bwd", "star_cif_quantity_fwd",
                 "star_cif_resize", "star_cif_fire_scan", "star_scorer_fwd", "star_scorer_bwd", "star_xattn_fwd",
                 "star_xattn_bwd", "star_xattn_dbias_sum", "star_type_embed_add", "star_type_embed_grad",
                 "star_type_embed_grad_workspace", "star_last_error", "star_version"):
        assert must in names


def test_library_exports_every_declared_symbol():
    from star_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "libstar_nugget.so must be built in-tree (python star_amd/csrc/build.py)"
    handle = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared_functions():
        assert hasattr(handle, name), f"{name} declared in the header but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature in star_amd/_lib.py"
    assert sorted(_lib.SIGNATURES) == declared_functions()
    assert _lib.lib().star_version() >= 100


def test_argument_errors_do_not_need_a_gpu():
    """Shape validation happens on the host before any launch and reports through star_last_error()."""
    from star_amd import _lib
    L = _lib.lib()
    rc = L.star_topk_select(None, None, None, None, 0.25, 1, 8, 2, None, None, None, None, None, None, None)
    assert rc == -1 and b"star_topk_select" in L.star_last_error()
    rc = L.star_scorer_fwd(None, None, None, None, None, 10, 100, None, None, None, 0, None)
    assert rc == -1
    assert L.star_scorer_fwd_workspace(48000, 1024) == 4 * 48128 * 4   # 4 column tiles x rows padded to 256
    assert L.star_scorer_relu_bits_bytes(48000, 1024) == 48128 * 128     # one bit per element of H
    assert L.star_scorer_relu_bits_bytes(300, 384) == 2 * 2 * 8192         # column tiles padded to 256


def test_ops_refuse_cpu_tensors():
    import torch
    from star_amd import ops
    with pytest.raises(ops.StarDeviceError):
        ops.cif_resize(torch.rand(2, 8), torch.tensor([2, 3]))
    with pytest.raises(ops.StarDeviceError):
        ops.gather_rows(torch.rand(2, 8, 4), torch.zeros(2, 3, dtype=torch.long))
