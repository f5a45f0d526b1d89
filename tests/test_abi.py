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
    # ... and nothing else: the library is built with hidden visibility, its internal launchers (utx_launch_*, *_impl) are not ABI
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted({l.split()[-1] for l in out.splitlines() if " T " in l and l.split()[-1].startswith("utx_")})
    assert exported == names, "exported but not declared: %s; declared but not exported: %s" % (sorted(set(exported) - set(names)), sorted(set(names) - set(exported)))


def test_product_library_has_no_wrong_result_switches():
    """the timing ablations that compute wrong results (UTX_ATTN_VAR / UTX_ATTN_DEBUG / UTX_GEMM_DEBUG) are compiled only into
    libunitex_hip_ablate.so; the product library refuses them by name and lists only result-preserving options."""
    import ctypes as C
    import pytest
    lib = _lib.load_library()
    assert lib.utx_is_ablation_build() == 0
    for n in ("UTX_ATTN_VAR", "UTX_ATTN_DEBUG", "UTX_GEMM_DEBUG"):
        assert lib.utx_set_option(n.encode(), 3) == -7
        v = C.c_int()
        assert lib.utx_get_option(n.encode(), C.byref(v)) == -7
        with pytest.raises(ValueError):
            _lib.set_option(n, 1)
    assert lib.utx_set_option(b"UTX_NO_SUCH", 1) == -2
    opts = _lib.get_options()
    assert set(opts) == set(_lib.OPTION_NAMES)
    # the binding's list IS the library's table (capi.cpp kOptions minus the ablation-only names): the conftest guard that fails a test which leaves an option changed
    # sees exactly the options it can read -- a name missing here would be a switch tests could leak unnoticed
    import re
    table = re.findall(r'\{"(UTX_[A-Z0-9_]+)",\s*&UtxOptions::\w+,\s*(true|false)\}', open(os.path.join(ROOT, "unitex_amd", "csrc", "capi.cpp")).read())
    assert len(table) >= 20 and {n for n, abl in table if abl == "false"} == set(_lib.OPTION_NAMES), sorted({n for n, abl in table if abl == "false"} ^ set(_lib.OPTION_NAMES))
    _lib.set_option("UTX_GEMM_TILE", 128)
    assert _lib.get_options()["UTX_GEMM_TILE"] == 128
    _lib.set_option("UTX_GEMM_TILE", 0)
    # no getenv on the launch path
    for f in os.listdir(os.path.join(ROOT, "unitex_amd", "csrc")):
        if f.endswith(".hip"):
            assert "getenv" not in open(os.path.join(ROOT, "unitex_amd", "csrc", f)).read(), f


def test_struct_layouts_match():
    assert _lib.check_abi()


def test_no_gpu_means_loud_failure():
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        _lib.Context(0)


def test_bit_exact_kernels_are_built_without_fp_contraction():
    """The float32 outputs that the GPU tests compare BIT FOR BIT with the C / numpy oracle (clip transform, barycentrics, interpolation, gathered colours,
    LBVH boxes, pull-push, k-NN distances, LayerNorm / RoPE roundings) are bit-exact only because both sides evaluate the same expression order
    without fused multiply-adds: the build table must carry -ffp-contract=off for exactly those sources, and the oracle's Makefile as well.  The files
    outside the set are the ones whose tests carry a tolerance (MFMA GEMMs / attention: bf16 bounds; vae.hip: bf16 bounds) or compute integers only
    (unwrap.hip: label propagation)."""
    import re
    from unitex_amd.csrc import build as b
    flags = {name: extra for name, extra in b.SOURCES}
    must = ["dit_elementwise.hip", "raster.hip", "bvh.hip", "backproject.hip", "texture_post.hip", "knn.hip"]
    for name in must:
        assert "-ffp-contract=off" in flags[name], "%s must be compiled with -ffp-contract=off (bit-exact tests depend on it)" % name
    for name in ("gemm.hip", "gemm_w4.hip", "gemm_pers.hip", "attention.hip", "attention_glds.hip", "attention_q64.hip", "vae.hip", "unwrap.hip"):
        assert "-ffp-contract=off" not in flags[name]
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(here, "unitex_amd", "csrc", "unwrap.hip")).read()
    body = re.sub(r"//[^\n]*", "", src)
    assert not re.search(r"\bfloat\b|\bdouble\b", body), "unwrap.hip is outside the no-contraction set because it has no floating-point arithmetic"
    mk = open(os.path.join(here, "oracle", "Makefile")).read()
    assert "-ffp-contract=off" in mk, "the C oracle must be built without contraction as well"


def test_bvh_workspace_build_refuses_a_bad_workspace_before_touching_the_device():
    """utx_bvh_build_ws (round 5): the caller provides every byte of the tree.  Size query monotone in the face count; a misaligned or short workspace is refused with -2
    before any device call (so this runs without a GPU)."""
    import ctypes as C
    from unitex_amd import _lib
    lib = _lib.load_library()
    sizes = [int(lib.utx_bvh_workspace_bytes(F)) for F in (0, 1, 2, 1000, 50000, 200000)]
    assert sizes[0] == 0 and all(a <= b for a, b in zip(sizes[1:], sizes[2:])) and sizes[4] >= 50000 * (2 * 3 * 4 + 2 * 6 * 4 + 6 * 4 + 16 + 2 * 4 + 4 + 2 * 32 + 48)
    buf = (C.c_char * 8192)()
    base = (C.addressof(buf) + 255) & ~255
    v, f, h = (C.c_float * 9)(), (C.c_int * 3)(0, 1, 2), C.c_void_p()
    assert lib.utx_bvh_build_ws(None, v, 3, f, 1, C.c_void_p(base | 16), C.c_size_t(4096), C.byref(h), None) == -2      # not 256-byte aligned
    assert lib.utx_bvh_build_ws(None, v, 3, f, 1, C.c_void_p(base), C.c_size_t(sizes[1] - 1), C.byref(h), None) == -2   # one byte short
    assert lib.utx_bvh_build_ws(None, v, 3, f, 0, C.c_void_p(base), C.c_size_t(4096), C.byref(h), None) == -2           # no faces
    assert lib.utx_bvh_build_ws(None, v, 3, f, 1, None, C.c_size_t(4096), C.byref(h), None) == -2                        # no workspace
    assert not h.value
