"""The C-ABI library builds for gfx950, loads without a GPU and exports every symbol declared in
include/deepipr_hip.h (no compute calls here)."""
import ctypes
import os
import re

from deepipr_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, 'include', 'deepipr_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(deepipr_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    names = _declared()
    assert len(names) >= 16
    handle = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(handle, n), n
    assert sorted(_lib.SIGNATURES) == names          # the ctypes table mirrors the header one to one


def test_abi_version_and_error_string():
    handle = _lib.lib()
    assert handle.deepipr_abi_version() == _lib.ABI_VERSION
    # argument validation happens before any HIP call, so it is safe without a GPU
    rc = handle.deepipr_affine_relu_fwd(None, None, None, None, 1, 1, 1, 1, None)
    assert rc == -1
    assert b'affine_relu_fwd' in handle.deepipr_last_error()
    assert handle.deepipr_affine_relu_bwd_workspace_bytes(128, 512, 16) >= 2 * 512 * 8
    assert handle.deepipr_affine_relu_bwd_workspace_bytes(0, 512, 16) == 0


def test_missing_library_is_loud(monkeypatch, tmp_path):
    import pytest
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', str(tmp_path / 'nope.so'))
    with pytest.raises(_lib.HipLibraryMissing, match='no CPU or PyTorch fallback'):
        _lib.lib()
