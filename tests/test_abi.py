"""CPU-only: libswn_hip.so loads and exports exactly the entry points include/swn.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "swn.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(swn_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    """Both builds (bf16: libswn_hip.so, fp16: libswn_hip_f16.so) export exactly the declared entry points and say which 16-bit
    compute type they carry."""
    from switch_nerf_amd import _lib
    if not (os.path.exists(_lib.LIB_PATH) and os.path.exists(_lib.LIB_PATH_F16)):
        import __graft_entry__
        __graft_entry__.build()
    syms = header_symbols()
    assert len(syms) >= 20
    declared = set(syms)
    bound = set(_lib.SIGNATURES) | {"swn_last_error", "swn_route_workspace_bytes", "swn_chain_mask_words", "swn_gate_bwd_scratch_floats", "swn_wgrad_multi_workspace_bytes", "swn_heads_bwd_workspace_bytes",
                                    "swn_half_dtype"}
    assert bound == declared, (sorted(bound - declared), sorted(declared - bound))
    for path, half in ((_lib.LIB_PATH, _lib.BF16), (_lib.LIB_PATH_F16, _lib.F16)):
        lib = ctypes.CDLL(path)
        for s in syms:
            assert hasattr(lib, s), f"{s} declared in include/swn.h but not exported by {os.path.basename(path)}"
        lib.swn_version.restype = ctypes.c_int
        lib.swn_half_dtype.restype = ctypes.c_int
        assert lib.swn_version() >= 2 and lib.swn_half_dtype() == half


def test_half_build_selection():
    from switch_nerf_amd import _lib
    try:
        assert _lib.use_half("f16").swn_half_dtype() == _lib.F16 and _lib.half_kind() == "f16"
        rc = _lib.load().swn_mlp_chain(None, None)
        assert rc != 0
    finally:
        assert _lib.use_half("bf16").swn_half_dtype() == _lib.BF16 and _lib.half_kind() == "bf16"


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from switch_nerf_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "_libs", {})
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load()


def test_error_reporting_without_gpu():
    """Argument validation happens before any launch, so it can be exercised on a CPU-only box."""
    from switch_nerf_amd import _lib
    lib = _lib.load()
    rc = lib.swn_mlp_chain(None, None)
    assert rc != 0 and b"null descriptor" in lib.swn_last_error()
    rc = lib.swn_route_top1(None, None, None, 10, 3, 8, 1, 1, None, None, None, None, None, None, 0, None)
    assert rc != 0 and b"null pointer" in lib.swn_last_error()
