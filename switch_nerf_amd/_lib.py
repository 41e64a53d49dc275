"""ctypes binding of libswn_hip.so (C ABI: include/swn.h).  Fails loudly when the library is missing."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SWN_LIB") or os.path.join(_HERE, "libswn_hip.so")      # (SWN_LIB: A/B builds of the library, experiments only)
LIB_PATH_F16 = os.path.join(_HERE, "libswn_hip_f16.so")    # the same sources built with IEEE half as the 16-bit compute type

F32, BF16, F16 = 0, 1, 2


def source_hash() -> str:
    """sha256 over the kernel sources (csrc/*, include/swn.h, build.sh: names and contents, sorted) - the identity of the kernels a
    measurement ran.  profiles/traffic.json records the hash of the build its counter passes ran; bench.py reports HBM traffic from
    the table only when the sources it runs hash the same."""
    import hashlib
    h = hashlib.sha256()
    files = [os.path.join(_HERE, "csrc", f) for f in sorted(os.listdir(os.path.join(_HERE, "csrc")))]
    files += [os.path.join(os.path.dirname(_HERE), "include", "swn.h"), os.path.join(_HERE, "build.sh")]
    for f in files:
        if os.path.isfile(f):
            h.update(os.path.basename(f).encode() + b"\0")
            h.update(open(f, "rb").read())
    return h.hexdigest()

vp, i32, i64, f32, sz = C.c_void_p, C.c_int, C.c_long, C.c_float, C.c_size_t


class ChainLayer(C.Structure):
    _fields_ = [("w", vp), ("b", vp), ("save", vp), ("mask", vp), ("rowbias", vp), ("rows_per_bias", i32),
                ("n", i32), ("k", i32), ("relu", i32), ("skip", i32)]


class WgradItem(C.Structure):
    _fields_ = [("a", vp), ("b", vp), ("a_gather", vp), ("b_gather", vp), ("dw", vp), ("db", vp)]


class WgradJob(C.Structure):
    _fields_ = [("a", vp), ("b", vp), ("a_gather", vp), ("b_gather", vp), ("dw", vp), ("db", vp), ("dw_set_stride", sz), ("db_set_stride", sz),
                ("m_dim", i32), ("n_dim", i32), ("lda", i32), ("ldb", i32), ("ldw", i32)]


class PackItem(C.Structure):
    _fields_ = [("master", vp), ("out", vp), ("n_wsets", i32), ("in_dim", i32), ("out_dim", i32), ("transpose", i32), ("in_rows", i32), ("out_cols", i32)]


class HashCfg(C.Structure):
    _fields_ = [("n_levels", i32), ("log2_table", i32), ("base_res", i32), ("per_level_scale", f32), ("aabb_lo", f32 * 3),
                ("aabb_hi", f32 * 3)]


class ChainDesc(C.Structure):
    _fields_ = [("dtype", i32), ("n_layers", i32), ("n_groups", i32), ("n_wsets", i32), ("group_stride", i32),
                ("group_rows", vp), ("group_rows_clamp", i32), ("group_begin", vp), ("x", vp), ("x_gather", vp), ("x_save", vp), ("x_scale", vp), ("x_relu", i32),
                ("y", vp), ("y_add", vp), ("y_add_gather", vp), ("geometry", i32), ("tag", i32),
                ("comb_y", vp), ("comb_dsig", vp), ("comb_wsig", vp), ("comb_gate", vp), ("comb_dgate", vp), ("comb_dwsig", vp), ("comb_dwsig_ws", vp),
                ("heads_ws", vp), ("heads_bs", vp), ("heads_wc", vp), ("heads_bc", vp), ("heads_noise", vp), ("heads_raw", vp), ("sched", vp), ("x_features", i32),
                ("head_layers", i32), ("tail_first", i32), ("y_features", i32), ("tail_gate", vp), ("tail_dropped", vp), ("tail_n_dropped", vp),
                ("tail_dropped_max", i32), ("tail_tokens", i32), ("tail_bias_row", vp),
                ("layers", ChainLayer * 12)]


# name -> argtypes; every symbol declared in include/swn.h must be listed here (tests/test_abi.py checks both ways)
SIGNATURES = {
    "swn_version": [],
    "swn_mfma_probe": [vp, vp],
    "swn_sample_pe": [vp, vp, vp, f32, i32, i32, i32, i32, i32, vp, vp, i32, vp, i32, vp],
    "swn_pe_from_z": [vp, vp, i32, i32, i32, i32, vp, i32, vp],
    "swn_sample_pdf": [vp, vp, vp, i32, i32, i32, vp, vp],
    "swn_merge_samples": [vp, vp, vp, vp, i32, i32, i32, vp, vp, vp, vp],
    "swn_unmerge_grad": [vp, vp, i32, i32, i32, vp, vp, vp],
    "swn_gate_fwd": [vp, i32, vp, vp, vp, i32, i32, i32, vp, vp, vp, vp, vp],
    "swn_gate_fwd_noise": [vp, i32, vp, vp, vp, vp, f32, i32, i32, i32, vp, vp, vp, vp, vp],
    "swn_gate_bwd": [vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp],
    "swn_gate_bwd_dense": [vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp],
    "swn_gate_logits": [vp, i32, vp, vp, f32, i32, i32, i32, vp, vp],
    "swn_load_importance_fwd": [vp, vp, vp, f32, i32, i32, vp, vp, vp, vp],
    "swn_load_importance_bwd": [vp, vp, vp, vp, vp, f32, i32, i32, vp, vp],
    "swn_topk_select": [vp, i32, i32, i32, vp, vp, vp, vp],
    "swn_route_topk": [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, sz, vp],
    "swn_topk_gate_bwd": [vp, vp, vp, i32, i32, i32, vp, vp],
    "swn_dispatch_fwd_more": [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp],
    "swn_dispatch_bwd_data_more": [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp],
    "swn_route_top1": [vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, sz, vp],
    "swn_route_top1x": [vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp, sz, vp],
    "swn_dispatch_fwd": [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp],
    "swn_dispatch_bwd_data": [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp],
    "swn_dispatch_bwd_gate": [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp],
    "swn_dispatch_nobatch_fwd": [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i64, vp],
    "swn_dispatch_nobatch_bwd_data": [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp],
    "swn_dispatch_nobatch_bwd_gate": [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp],
    "swn_route_pack": [vp, vp, vp, i32, i32, i32, vp, vp, vp, vp],
    "swn_route_dropped": [vp, vp, vp, i32, i32, i32, i32, vp, vp, vp],
    "swn_combine_fwd": [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp],
    "swn_combine_bwd": [vp, vp, vp, vp, vp, i32, i32, i32, vp, vp, vp],
    "swn_heads_fwd": [vp, vp, i32, vp, vp, vp, vp, vp, i32, i32, i32, vp, vp],
    "swn_heads_bwd": [vp, vp, i32, vp, vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, i32, vp, vp, sz, vp],
    "swn_group_colsum": [vp, i32, i32, i32, i32, vp, vp],
    "swn_composite_fwd": [vp, vp, f32, f32, i32, i32, vp, vp, vp, vp, vp],
    "swn_composite_bwd": [vp, vp, f32, f32, vp, i32, i32, vp, vp],
    "swn_sample_z": [vp, vp, vp, f32, i32, i32, vp, vp],
    "swn_mip_encode": [vp, vp, vp, i32, i32, i32, i32, vp, i32, vp],
    "swn_mip_resample": [vp, vp, vp, f32, i32, i32, i32, vp, vp],
    "swn_fg_bounds": [vp, vp, vp, i32, vp, vp, vp, vp, vp, vp],
    "swn_bg_sample_pe": [vp, vp, vp, vp, vp, f32, i32, i32, i32, i32, vp, vp, vp, vp, i32, vp],
    "swn_composite_bounded_fwd": [vp, vp, vp, i32, vp, i32, i32, vp, vp, vp, vp, vp, vp],
    "swn_composite_bounded_bwd": [vp, vp, vp, i32, vp, vp, i32, i32, vp, vp],
    "swn_hash_encode_fwd": [vp, vp, i32, i32, C.POINTER(HashCfg), vp, i32, vp, i32, vp],
    "swn_hash_encode_bwd": [vp, vp, i32, i32, C.POINTER(HashCfg), vp, i32, i32, vp, vp],
    "swn_hash_encode_bwd_xcd": [vp, vp, i32, i32, C.POINTER(HashCfg), vp, i32, i32, vp, vp, vp],
    "swn_hash_encode_bwd_binned": [vp, vp, i32, i32, C.POINTER(HashCfg), vp, i32, i32, vp, vp, sz, vp],
    "swn_gather_rows": [vp, vp, i64, i32, vp, vp],
    "swn_sign_bits_pack": [vp, i64, i32, vp, vp],
    "swn_sign_bits_unpack": [vp, i64, i32, vp, vp],
    "swn_scatter_rows": [vp, vp, i64, i32, vp, vp],
    "swn_owner_aux": [vp, vp, vp, i64, i32, i32, i32, vp, vp],
    "swn_owner_aux_split": [vp, i64, vp, vp, vp, vp],
    "swn_ray_bias_grad_bits": [vp, vp, vp, vp, i32, i32, i32, vp, vp],
    "swn_mlp_chain": [C.POINTER(ChainDesc), vp],
    "swn_chain_big_ok": [C.POINTER(ChainDesc)],
    "swn_pack_weights": [vp, vp, i32, i32, i32, i32, i32, vp],
    "swn_pack_weights_batched": [C.POINTER(PackItem), i32, i32, vp],
    "swn_chain_tile_rows": [i32],
    "swn_wgrad_blocks": [C.POINTER(WgradItem), i32, i32, i32, i32, i32, i32, i32, sz, sz, i32, i32, i32, vp, i32, i32, i32, vp, sz, vp],
    "swn_wgrad_batched": [C.POINTER(WgradItem), i32, i32, i32, i32, i32, i32, i32, vp, i32, i32, i32, vp, sz, vp],
    "swn_wgrad_multi": [C.POINTER(WgradJob), i32, i32, i32, i32, i32, vp, i32, vp, i32, vp, sz, vp],
    "swn_wgrad": [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, i32, vp, vp, i32, i32, vp, sz, vp],
    "swn_ray_feat_fwd": [vp, i32, i32, i32, vp, i32, vp, i32, vp, vp, i32, i32, vp, vp, vp],
    "swn_step_loss": [vp, vp, i32, vp, i32, vp, i32, f32, vp, vp, vp, vp, vp, vp],
    "swn_emb_grad": [vp, i32, vp, i32, i32, i32, i32, vp, vp],
    "swn_ray_feat_wgrad": [vp, vp, i32, i32, i32, vp, vp, vp, sz, vp],
    "swn_adam_step": [vp, vp, vp, vp, vp, i32, i64, f32, f32, f32, f32, i32, f32, vp],
    "swn_cast": [vp, vp, i32, i64, vp],
    "swn_cast_transpose": [vp, vp, i32, i32, i32, i32, vp],
}

_lib = None
_libs = {}            # "bf16" / "f16" -> loaded library
_half = "bf16"        # which build the process currently talks to


def use_half(kind: str):
    """Select the build of the library by its 16-bit compute type: "bf16" (libswn_hip.so, default) or "f16" (libswn_hip_f16.so).
    Both carry the same entry points and both compute fp32; a model of compute dtype torch.float16 selects "f16" when it is built
    and checks the selection on every step (the two 16-bit types share one dtype slot per build)."""
    global _half, _lib
    assert kind in ("bf16", "f16")
    _half = kind
    _lib = _libs.get(kind)
    return load()


def half_kind() -> str:
    return _half


def load():
    """Load the HIP library (once per build).  No fallback: a missing library is an error."""
    global _lib
    if _lib is not None:
        return _lib
    path = LIB_PATH if _half == "bf16" else LIB_PATH_F16
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(switch_nerf_amd/build.sh).  There is no CPU fallback for the hot path.")
    lib = C.CDLL(path)
    lib.swn_last_error.restype = C.c_char_p
    lib.swn_last_error.argtypes = []
    lib.swn_route_workspace_bytes.restype = sz
    lib.swn_route_workspace_bytes.argtypes = [i32, i32, i32]
    lib.swn_route_sync_bytes.restype = sz
    lib.swn_route_sync_bytes.argtypes = []
    lib.swn_gate_bwd_scratch_floats.restype = sz
    lib.swn_gate_bwd_scratch_floats.argtypes = [i32, i32, i32]
    lib.swn_load_importance_workspace_floats.restype = sz
    lib.swn_load_importance_workspace_floats.argtypes = [i32, i32]
    lib.swn_wgrad_multi_workspace_bytes.restype = sz
    lib.swn_wgrad_multi_workspace_bytes.argtypes = [i32, i32]
    lib.swn_heads_bwd_workspace_bytes.restype = sz
    lib.swn_heads_bwd_workspace_bytes.argtypes = [i32, i32, i32]
    lib.swn_ray_feat_wgrad_workspace_bytes.restype = sz
    lib.swn_ray_feat_wgrad_workspace_bytes.argtypes = [i32, i32, i32]
    lib.swn_chain_dwsig_workspace_bytes.restype = sz
    lib.swn_chain_dwsig_workspace_bytes.argtypes = [i32, i32]
    lib.swn_hash_bwd_workspace_bytes.restype = sz
    lib.swn_hash_bwd_workspace_bytes.argtypes = [i64, C.POINTER(HashCfg)]
    lib.swn_chain_mask_words.restype = i64
    lib.swn_chain_mask_words.argtypes = [i32, i32, i32, i32]
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = i32
        fn.argtypes = args
    lib.swn_half_dtype.restype = i32
    lib.swn_half_dtype.argtypes = []
    assert lib.swn_half_dtype() == (BF16 if _half == "bf16" else F16), f"{path} is not the {_half} build"
    _lib = _libs[_half] = lib
    return lib


def call(name, *args):
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise RuntimeError(f"{name} failed: {lib.swn_last_error().decode()}")
