"""CPU-side checks of the C-ABI boundary: the library builds/loads without a GPU and exports every
symbol include/unitex_hip.h declares; ctypes struct layouts match the C structs."""
import os
import re

from unitex_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "unitex_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(utx_[a-z0-9_]+)\s*\(", src)))


def test_library_loads_and_exports_all_declared_symbols():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = _lib.load_library()
    names = _declared_symbols()
    assert len(names) >= 10
    for n in names:
        assert hasattr(lib, n), "declared in unitex_hip.h but not exported: %s" % n
        assert n in _lib.SYMBOLS, "declared in unitex_hip.h but not bound in _lib.SYMBOLS: %s" % n
    assert lib.utx_version() == 100


def test_struct_layouts_match():
    assert _lib.check_abi()


def test_no_gpu_means_loud_failure():
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        _lib.Context(0)
