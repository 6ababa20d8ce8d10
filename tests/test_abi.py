"""CPU: the C-ABI library builds for gfx950, loads without a GPU and exports exactly the entry points
declared in include/rf_hip.h; the ctypes signature table matches the header's parameter lists."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "rf_hip.h")


def _declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(?:int64_t|int|float|const char\*)\s+(rf_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        args = [a.strip() for a in m.group(2).replace("\n", " ").split(",") if a.strip() and a.strip() != "void"]
        out[m.group(1)] = args
    return out


def _ctype_of(arg: str):
    if "*" in arg:
        return ctypes.c_void_p
    if arg.startswith("int64_t"):
        return ctypes.c_int64
    if arg.startswith("float"):
        return ctypes.c_float
    if arg.startswith("int"):
        return ctypes.c_int
    raise AssertionError(arg)


def test_library_exports_every_declared_symbol():
    from routeformer_amd import _hip
    if not os.path.exists(_hip.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = _hip.lib()
    decl = _declared()
    assert len(decl) >= 20
    for name in decl:
        assert hasattr(lib, name), f"{name} declared in rf_hip.h but not exported"
    assert lib.rf_version() >= 1


def test_ctypes_signatures_match_header():
    from routeformer_amd import _hip
    decl = _declared()
    for name, argtypes in _hip.SIGNATURES.items():
        assert name in decl, name
        want = [_ctype_of(a) for a in decl[name]]
        assert want == argtypes, (name, decl[name])
    missing = set(decl) - set(_hip.SIGNATURES) - {"rf_last_error"}
    assert not missing, missing


def test_argument_validation_without_gpu():
    """Bad arguments are rejected before any launch (safe to call on a GPU-less box)."""
    from routeformer_amd import _hip
    lib = _hip.lib()
    assert lib.rf_gemm(None, 1, 1, None, 1, 1, None, 1, 4, 4, 4, None, None, 0, 0, 0, 0, None, 0, None, 0, 0, 0, 1,
                       None, 0, None, None, None) == -1
    assert b"invalid argument" in lib.rf_last_error()
    assert lib.rf_layernorm_bwd_parts(12480) == 768 and lib.rf_colsum_parts(12480, 128) == 49


def test_missing_library_fails_loudly(monkeypatch):
    from routeformer_amd import _hip
    monkeypatch.setattr(_hip, "_lib", None)
    monkeypatch.setattr(_hip, "LIB_PATH", "/nonexistent/librf_hip.so")
    with pytest.raises(_hip.HipLibraryError):
        _hip.lib()


def test_split_counts_without_gpu():
    """The pure slab-count helpers of the split-K products (no launch): ``rf_gemm_split_count`` is rf_gemm's own clamping
    of ``splitk`` (64-wide K granules, no empty slice), ``rf_gemm_skinny_split`` the skinny kernel's K slices of 1 024 --
    what a caller sizes the slab workspace of ``rf_gemm_partials`` / ``rf_gemm_skinny_partials`` with."""
    from routeformer_amd import _hip
    lib = _hip.lib()

    def mirror(K, s):
        ktiles = -(-K // 64)
        s = max(1, min(s, ktiles))
        kchunk = -(-ktiles // s) * 64
        return -(-K // kchunk)

    for K in (1, 63, 64, 100, 128, 207, 208, 832, 2496, 3328, 4097):
        for s in (1, 2, 3, 5, 8, 13, 16, 64):
            got = lib.rf_gemm_split_count(K, s)
            assert got == mirror(K, s) and 1 <= got <= s, (K, s, got)
    assert lib.rf_gemm_split_count(3328, 16) == 13 and lib.rf_gemm_split_count(832, 3) == 3
    # skinny: only layouts are inspected, never the pointers' contents (any aligned non-null address will do)
    a = ctypes.c_void_p(0x1000)
    assert lib.rf_gemm_skinny_split(a, 832, 1, a, 1, 832, 40, 832, 832) == 1
    assert lib.rf_gemm_skinny_split(a, 3328, 1, a, 1, 3328, 40, 832, 3328) == 4
    assert lib.rf_gemm_skinny_split(a, 3328, 1, a, 1, 3328, 700, 832, 3328) == 0   # M > 640: the tiled kernel
    assert lib.rf_gemm_skinny_split(None, 832, 1, a, 1, 832, 40, 832, 832) == 0
