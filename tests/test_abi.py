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
    bound = set(_lib.SIGNATURES) | {"swn_last_error", "swn_route_workspace_bytes", "swn_route_sync_bytes", "swn_chain_mask_words", "swn_gate_bwd_scratch_floats", "swn_wgrad_multi_workspace_bytes", "swn_heads_bwd_workspace_bytes", "swn_ray_feat_wgrad_workspace_bytes", "swn_chain_dwsig_workspace_bytes", "swn_hash_bwd_workspace_bytes", "swn_load_importance_workspace_floats",
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
    rc = lib.swn_route_top1x(None, None, None, 10, 3, 8, 1, 1, None, None, None, None, None, None, None, None, 1, None, 0, None)
    assert rc != 0 and b"swn_route_top1x: null pointer" in lib.swn_last_error()
    assert lib.swn_route_sync_bytes() == 4096


def test_round3_entry_points_validate_before_launch():
    """The entry points added in round 3 (ordered heads backward with its workspace / per-ray sums, ordered embedding gradient, the
    balanced weight-gradient launch, the heads fused into the tail forward chain) reject bad arguments with a message and launch
    nothing - exercised without a GPU (fake non-null pointers are never dereferenced on the host)."""
    import ctypes as C
    from switch_nerf_amd import _lib
    lib = _lib.load()
    p = C.c_void_p(0x1000)                                   # any non-null address: validation only
    err = lambda: lib.swn_last_error().decode()
    # heads backward: workspace size, rows_per_group must divide the point count and come with an output
    need = lib.swn_heads_bwd_workspace_bytes(1000, 256, 128)
    assert need >= (256 + 3 * 128 + 4) * 4
    args = [p, p, _lib.F32, p, p, p, 1000, 256, 128, p, p, p, p, p, p]
    assert lib.swn_heads_bwd(*args, 0, None, p, need - 4, None) != 0 and "workspace" in err()
    assert lib.swn_heads_bwd(*args, 7, p, p, need, None) != 0 and "rows_per_group" in err()
    assert lib.swn_heads_bwd(*args, 10, None, p, need, None) != 0 and "rows_per_group" in err()
    assert lib.swn_heads_bwd(*args, 0, None, None, need, None) != 0 and "null pointer" in err()
    # embedding gradient
    assert lib.swn_emb_grad(p, 48, p, 1, 100, 48, 0, p, None) != 0 and "bad sizes" in err()
    assert lib.swn_emb_grad(p, 40, p, 1, 100, 48, 10, p, None) != 0 and "bad sizes" in err()
    assert lib.swn_emb_grad(None, 48, p, 1, 100, 48, 10, p, None) != 0 and "null pointer" in err()
    assert lib.swn_emb_grad(p, 48, p, 1, 0, 48, 10, p, None) == 0            # no rays: nothing to do, nothing launched
    # balanced weight-gradient launch: job count, workspace
    assert lib.swn_wgrad_multi_workspace_bytes(7, 8) > 0
    jobs = (_lib.WgradJob * 1)()
    assert lib.swn_wgrad_multi(jobs, 0, _lib.BF16, 1, 1, 256, None, 256, None, 0, p, 1 << 30, None) != 0
    assert lib.swn_wgrad_multi(jobs, 9, _lib.BF16, 1, 1, 256, None, 256, None, 0, p, 1 << 30, None) != 0
    # fused heads: only on the 64-row kernels with tag 4, with all four parameter arrays, rows of at most 1 KiB
    d = _lib.ChainDesc()
    d.dtype, d.n_layers, d.n_groups, d.n_wsets, d.group_stride, d.group_rows_clamp = _lib.BF16, 2, 1, 1, 640, 640
    d.x, d.y = p, None
    for i, (n, k) in enumerate(((256, 256), (128, 256))):
        d.layers[i].w, d.layers[i].n, d.layers[i].k = p, n, k
    assert lib.swn_mlp_chain(C.byref(d), None) != 0 and "x / y" in err()      # no y and no fused heads
    d.heads_raw = p
    assert lib.swn_mlp_chain(C.byref(d), None) != 0 and "fused heads need" in err()
    d.heads_ws = d.heads_bs = d.heads_wc = d.heads_bc = p
    d.tag = 3
    assert lib.swn_mlp_chain(C.byref(d), None) != 0 and "tag 4" in err()
    d.tag, d.dtype = 4, _lib.F32
    d.layers[0].n = d.layers[0].k = d.layers[1].k = 512
    assert lib.swn_mlp_chain(C.byref(d), None) != 0 and "1 KiB" in err()


def test_fused_tail_descriptors_validate_before_launch():
    """swn_chain_desc.tail_first / head_layers (round 4: the dense tail inside the expert launches) are tied to geometry 7 and their own
    tags, and a descriptor that the persistent kernel cannot run (a relu on the gate layer, a missing drop list, a token count whose
    rows do not fit 32-bit offsets) is rejected with a message - nothing is launched (no GPU needed)."""
    import ctypes as C
    from switch_nerf_amd import _lib
    lib = _lib.load()
    p = C.c_void_p(0x1000)
    err = lambda: lib.swn_last_error().decode()

    def desc(n_layers):
        d = _lib.ChainDesc()
        d.dtype, d.n_layers, d.n_groups, d.n_wsets, d.group_stride, d.group_rows_clamp = _lib.BF16, n_layers, 8, 8, 512, 512
        d.x, d.y, d.x_gather = p, p, p
        for i in range(n_layers):
            d.layers[i].w, d.layers[i].n, d.layers[i].k = p, 256, 256
        return d
    d = desc(9)
    d.tail_first, d.geometry, d.tag = 7, 1, 7
    assert lib.swn_mlp_chain(C.byref(d), None) != 0 and "geometry 7 with tag 7" in err()
    d.geometry, d.tag = 7, 1
    assert lib.swn_mlp_chain(C.byref(d), None) != 0 and "geometry 7 with tag 7" in err()
    d.tag, d.tail_first = 7, 0
    assert lib.swn_mlp_chain(C.byref(d), None) != 0 and "tag 7 is the fused-tail" in err()
    d.tail_first = 7                                          # no gate values / drop list / token count yet
    assert lib.swn_mlp_chain(C.byref(d), None) != 0 and "a fused tail" in err()
    d.tail_gate = d.tail_dropped = d.tail_n_dropped = p
    d.tail_tokens, d.tail_dropped_max, d.y_features = 4096, 4096, 128
    d.layers[8].relu = 1
    d.layers[6].relu = 1                                      # the gate layer's ReLU comes with the scaling
    assert lib.swn_mlp_chain(C.byref(d), None) != 0 and "a fused tail" in err()
    d.layers[6].relu = 0
    d.tail_tokens = 1 << 23                                   # 2^23 rows of 512 bytes: past the 32-bit store offsets
    assert lib.swn_mlp_chain(C.byref(d), None) != 0 and "a fused tail" in err()
    # backward: head layers
    b = desc(9)
    b.head_layers, b.geometry, b.tag = 2, 7, 2
    assert lib.swn_mlp_chain(C.byref(b), None) != 0 and "geometry 7 with tag 8" in err()
    b.tag, b.head_layers = 8, 0
    assert lib.swn_mlp_chain(C.byref(b), None) != 0 and "tag 8 is the expert backward" in err()
    b.head_layers = 2                                         # no combine operands, no 128-feature input
    assert lib.swn_mlp_chain(C.byref(b), None) != 0 and "a fused tail" in err()
    # the sigma head's weight gradient from the fused backward launch: destination and workspace come together, with the combine operands
    b.comb_dwsig = p
    assert lib.swn_mlp_chain(C.byref(b), None) != 0 and "comb_dwsig and comb_dwsig_ws come together" in err()
    b.comb_dwsig_ws = p
    assert lib.swn_mlp_chain(C.byref(b), None) != 0 and "comb_dwsig and comb_dwsig_ws come together" in err()      # (no comb_y / comb_dsig)
    f = desc(2)
    f.comb_dwsig = f.comb_dwsig_ws = f.comb_y = f.comb_dsig = p                                                     # (no head layers)
    assert lib.swn_mlp_chain(C.byref(f), None) != 0 and "comb_dwsig and comb_dwsig_ws come together" in err()
    # its workspace: 8 wave slots of 256 floats per 256-row tile of every group + 1024 run sums
    assert lib.swn_chain_dwsig_workspace_bytes(128, 16384) == (128 * 64 * 8 + 1024) * 1024
    assert lib.swn_chain_dwsig_workspace_bytes(16, 500) == (16 * 2 * 8 + 1024) * 1024 and lib.swn_chain_dwsig_workspace_bytes(0, 500) == 0
