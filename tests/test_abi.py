"""CPU: the C-ABI library loads and exports every symbol include/aurppo.h declares; the product
fails loudly without its HIP extension / without a GPU (no silent fallback)."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "aurppo.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(aurppo_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_all_exported():
    import __graft_entry__ as g
    g.build()
    from aur_ppo_amd import _lib
    lib = _lib.load()
    decl = _declared()
    assert len(decl) >= 17
    assert sorted(_lib.SYMBOLS) == decl
    for name in decl:
        assert hasattr(lib, name), name
    assert lib.aurppo_version() == 1
    assert lib.aurppo_loss_workspace_bytes(1024) > 0 and lib.aurppo_clip_workspace_bytes(10) > 0


def test_argument_validation_without_gpu():
    from aur_ppo_amd import _lib
    lib = _lib.load()
    assert lib.aurppo_gae_f32(None, None, None, None, None, None, None, 4, 4, 0.99, 0.95, 0, None) == -1
    assert b"null pointer" in lib.aurppo_last_error()
    assert lib.aurppo_arange_i32(None, 4, None) == -1
    assert lib.aurppo_mt19937_destroy(None) == 0


def test_weight_gradient_workspace_plans_without_gpu():
    """aurppo_conv3x3_wgrad_ws_bytes / aurppo_linear_wgrad_ws_bytes are host-side plans (tile and slice counts): every hidden layer
    of both image configs gets a workspace of S x Co x Ci x 9 floats with S the slice count; unsupported shapes say 0."""
    from aur_ppo_amd import _lib
    lib = _lib.load()
    for B, Ci, Co, S, pad in [(8192, 16, 32, 64, 1), (8192, 32, 64, 32, 1), (8192, 64, 128, 16, 1), (8192, 128, 256, 8, 1),
                              (8192, 256, 256, 8, 0), (4096, 64, 128, 42, 1), (4096, 512, 512, 5, 0), (1, 3, 5, 3, 1)]:
        nb = lib.aurppo_conv3x3_wgrad_ws_bytes(B, Ci, Co, S, S, pad)
        per_slice = Co * Ci * 9 * 4
        assert nb >= per_slice + 64 and (nb - 64) % per_slice == 0, (B, Ci, Co, S, pad, nb)
        tiles = -(-Co // (128 if Co > 64 else 64)) * -(-(Ci * 9) // (128 if Co > 64 else 256))
        slices = (nb - 64) // per_slice
        assert 1 <= slices and slices * tiles <= 4 * 512, (slices, tiles)          # at most four rounds of two workgroups per CU
    assert lib.aurppo_conv3x3_wgrad_ws_bytes(4, 16, 32, 8, 8, 3) == 0              # padding outside 0..2
    assert lib.aurppo_conv3x3_wgrad_ws_bytes(4, 16, 32, 2, 2, 0) == 0              # empty output
    assert lib.aurppo_conv3x3_wgrad_ws_bytes(1, 512, 8, 1024, 1024, 1) == 0        # one image of 2 GB
    assert lib.aurppo_linear_wgrad_ws_bytes(131072, 256, 256) >= 256 * 256 * 4 + 64
    assert lib.aurppo_linear_wgrad_ws_bytes(0, 256, 256) == 0
    assert lib.aurppo_conv3x3_wgrad_f32(None, None, None, 1, 16, 32, 8, 8, 1, None, None) == -1 and b"null pointer" in lib.aurppo_last_error()


@pytest.mark.skipif(torch.cuda.is_available(), reason="CPU-only check")
def test_no_cpu_fallback_exists():
    from aur_ppo_amd import hip_ops as H
    x = torch.zeros(4, 4)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        H.gae(x, x, x, x[0], x[0], 0.99, 0.95)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        H.MT19937(1, 16)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from aur_ppo_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.AurppoLibraryMissing, match="no CPU or PyTorch fallback"):
        _lib.load()


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "aur_ppo_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
