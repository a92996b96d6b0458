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
