"""ctypes binding of ``csrc/librf_hip.so`` (C ABI declared in ``include/rf_hip.h``).

There is NO fallback: if the shared library is missing or a call fails, this raises.  Tensors are
passed as raw device pointers; every call is asynchronous on the caller's current HIP stream.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_float, c_int, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# (RF_HIP_LIB: another build of the same library -- kernel experiments; the default is the in-tree build)
LIB_PATH = os.environ.get("RF_HIP_LIB") or os.path.join(_HERE, "csrc", "librf_hip.so")

_P, _I, _L, _F = c_void_p, c_int, c_int64, c_float

# name -> argtypes, mirroring include/rf_hip.h exactly (checked by tests/test_abi.py)
SIGNATURES = {
    "rf_gemm": [_P, _L, _L, _P, _L, _L, _P, _L, _I, _I, _I, _P, _P, _L, _I, _I, _I, _P, _L, _P, _L,
                _I, _I, _I, _P, _I, _P, _P, _P],
    "rf_gemm_skinny_split": [_P, _L, _L, _P, _L, _L, _I, _I, _I],
    "rf_gemm_skinny": [_P, _L, _L, _P, _L, _L, _P, _L, _I, _I, _I, _P, _P, _L, _I, _I, _I, _P, _L, _P, _L, _I, _P, _P],
    "rf_gemm_split_count": [_I, _I],
    "rf_gemm_partials": [_P, _L, _L, _P, _L, _L, _I, _I, _I, _I, _I, _P, _P],
    "rf_gemm_skinny_partials": [_P, _L, _L, _P, _L, _L, _I, _I, _I, _P, _P],
    "rf_colsum_parts": [_I, _I],
    "rf_colsum": [_P, _L, _I, _I, _P, _I, _P, _P],
    "rf_conv2d_nhwc": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _L, _L, _I, _I, _P],
    "rf_conv3x3_bf16_supported": [_I, _I],
    "rf_conv3x3_packed_elems": [_I, _I],
    "rf_conv3x3_pack_bf16": [_P, _P, _I, _I, _P],
    "rf_conv3x3_group_bf16": [_P, _I, _I, _P],
    "rf_conv3x3_bf16": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "rf_conv3x3_pair_supported": [_I, _I],
    "rf_conv3x3_pair_group_bf16": [_P, _I, _P],
    "rf_conv3x3s2_bf16_supported": [_I, _I, _I],
    "rf_conv3x3s2_packed_elems": [_I, _I],
    "rf_conv3x3s2_pack_bf16": [_P, _P, _I, _I, _P],
    "rf_conv3x3s2_bf16": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "rf_pointwise_bf16_supported": [_I, _I],
    "rf_pointwise_packed_elems": [_I, _I],
    "rf_pointwise_pack_bf16": [_P, _P, _I, _I, _P],
    "rf_pointwise_bf16": [_P, _P, _P, _P, _P, _L, _I, _I, _I, _P],
    "rf_stem_conv0": [_P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "rf_upsample_bilinear_nhwc": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _L, _I, _I, _P],
    "rf_add_relu": [_P, _P, _P, _I, _L, _I, _P],
    "rf_avgpool8_tokens": [_P, _I, _P, _I, _I, _I, _I, _P],
    "rf_unfold3_circular": [_P, _P, _I, _I, _I, _I, _P],
    "rf_fold3_circular": [_P, _P, _I, _I, _I, _I, _P],
    "rf_unfold3_circular_ld": [_P, _P, _I, _I, _I, _I, _I, _P],
    "rf_fold3_circular_ld": [_P, _P, _I, _I, _I, _I, _I, _P],
    "rf_layernorm_fwd": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _F, _P],
    "rf_layernorm_fwd_strided": [_P, _I, _L, _L, _P, _P, _P, _P, _P, _P, _I, _I, _F, _P],
    "rf_layernorm_fwd_slabs": [_P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _F, _P],
    "rf_layernorm_fwd_slabs_unfold": [_P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _P],
    "rf_layernorm_bwd_fold": [_P, _P, _P, _P, _P, _P, _P, _I, _P, _I, _I, _I, _P],
    "rf_layernorm_bwd_slabs": [_P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _P, _I, _I, _P],
    "rf_layernorm_bwd_parts": [_I],
    "rf_layernorm_bwd": [_P, _P, _P, _P, _P, _P, _P, _I, _P, _I, _I, _P],
    "rf_bn_stats": [_P, _P, _P, _I, _I, _P, _P, _P, _F, _P],
    "rf_bn_elu_pool_fwd": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _P],
    "rf_bn_train_elu_pool_fwd": [_P, _P, _P, _P, _P, _P, _P, _P, _F, _P, _P, _I, _I, _I, _F, _P],
    "rf_bn_train_elu_pool_fwd_slabs": [_P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _F, _P, _P, _I, _I, _I, _F, _P],
    "rf_bn_elu_pool_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _I, _P],
    "rf_bn_elu_pool_bwd_slab_ok": [_I, _I],
    "rf_bn_elu_pool_bwd_slabs": [_P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _I, _P],
    "rf_wgrad_grouped": [_P, _I, _I, _P],
    "rf_wgrad_tr": [_P, _I, _P],
    "rf_rowblock_linear_supported": [_I, _I, _I],
    "rf_rowblock_linear": [_P, _L, _P, _P, _P, _L, _P, _L, _I, _I, _I, _P, _P, _P, _P, _F, _P],
    "rf_rowblock_ffn_ln": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _F, _P],
    "rf_rowblock_linear_nn_supported": [_I, _I, _I],
    "rf_rowblock_linear_nn": [_P, _L, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _P, _L, _I, _P, _L, _I, _I, _I, _P],
    "rf_assemble_streams_fwd": [_P, _P, _P, _I, _I, _I, _I, _P],
    "rf_assemble_streams_bwd": [_P, _P, _I, _I, _I, _I, _P],
    "rf_traj_head_fwd": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _F, _I, _F, _F, _L, _L, _L, _P],
    "rf_traj_head_bwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _F, _L, _L, _P],
    "rf_attn_fwd": [_P, _P, _P, _L, _L, _L, _P, _I, _P, _I, _L, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F, _P],
    "rf_attn_fwd_full_scores": [_I, _I, _I, _I, _I, _I, _I, _I],
    "rf_attn_bwd": [_P, _P, _P, _L, _L, _L, _P, _I, _P, _P, _P, _P, _L, _L, _L, _I, _I, _I, _I, _I,
                    _I, _I, _F, _P],
    "rf_attn_bwd_slabs": [_P, _P, _P, _L, _L, _L, _P, _I, _L, _I, _P, _P, _P, _P, _L, _L, _L, _I, _I, _I, _I, _I, _I, _I, _F, _P],
    "rf_gather_frames": [_P, _I, _P],
    "rf_resize_area": [_P, _P, _L, _I, _I, _I, _I, _P],
    "rf_frame_hash": [_P, _P, _L, _L, _P, _L, _P],
    "rf_cache_lookup": [_P, _I, _P, _P, _I, _P, _P, _P],
    "rf_cache_insert": [_P, _I, _P, _P, _I, _P, _I, _P, _P],
    "rf_seqlayer_pack": [_P, _I, _P],
    "rf_seqlayer_pack_blocks": [_I, _I],
    "rf_seqlayer_pack_table": [_P, _P, _I, _I, _P],
    "rf_seqlayer_supported": [_I, _I, _I, _I, _I, _I],
    "rf_seqlayer_pack_bytes": [_I],
    "rf_seqlayer_fwd": [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F, _F, _F, _P, _I, _P],
    "rf_seqlayer_bwd_pack_bytes": [_I],
    "rf_seqlayer_bwd": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _F, _P, _I, _P],
    "rf_rng_seed": [_P, _L, _L, _P],
    "rf_rng_advance": [_P, _P],
    "rf_dropout": [_P, _P, _L, _F, _P, _I, _P, _P, _P],
    "rf_attn_fwd_drop": [_P, _P, _P, _L, _L, _L, _P, _I, _P, _I, _L, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F, _F, _P, _I,
                         _P, _P],
    "rf_attn_bwd_drop": [_P, _P, _P, _L, _L, _L, _P, _I, _P, _P, _P, _P, _L, _L, _L, _I, _I, _I, _I, _I, _I, _I, _F, _F, _P,
                         _I, _P, _P],
    "rf_motion_input": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "rf_rotate_head": [_P, _P, _P, _I, _I, _I, _F, _P],
    "rf_kernel_timer_arm": [],
    "rf_kernel_timer_collect": [_P, _I],
    "rf_sumsq_parts": [_L],
    "rf_sumsq": [_P, _L, _P, _P],
    "rf_adamw_clip": [_P, _P, _P, _P, _L, _P, _I, _F, _F, _F, _F, _F, _F, _I, _F, _P],
    "rf_adamw_clip_dev": [_P, _P, _P, _P, _L, _P, _I, _P, _I, _P],
    "rf_rowchain_supported": [_I, _I, _I],
    "rf_rowchain_fwd": [_P, _I, _F, _P, _P],
    "rf_rowchain_bwd": [_P, _I, _F, _P, _P],
    "rf_enclayer_tile_supported": [_I, _I, _I],
    "rf_enclayer_tile_fwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _F, _P, _I, _P],
    "rf_enclayer_tile_bwd": [_P] * 19 + [_I, _I, _I, _I, _I, _P, _F, _P, _I, _P],
    "rf_median_windows": [_P, _P, _I, _I, _I, _I, _P],
    "rf_motion_diff": [_P, _P, _I, _I, _I, _F, _F, _P],
    "rf_time_table": [_P, _P, _P, _I, _I, _P],
    "rf_time_table_bwd": [_P, _P, _I, _I, _I, _P],
    "rf_timeline_scatter": [_P, _P, _P, _L, _I, _I, _I, _P],
    "rf_timeline_gather": [_P, _P, _P, _L, _I, _I, _I, _P],
    "rf_smart_tail_fwd": [_P, _P, _I, _I, _I, _I, _I, _P],
    "rf_smart_tail_bwd": [_P, _P, _P, _I, _I, _I, _I, _I, _P],
    "rf_pad_cols": [_P, _P, _I, _I, _I, _P],
    "rf_unpad_cols": [_P, _P, _I, _I, _I, _I, _P],
    "rf_fuse_upsample_sum": [_P, _I, _I, _P],
    "rf_concat_pool_tokens": [_P, _P, _P, _P, _I, _I, _P, _I, _P],
    "rf_comm_available": [],
    "rf_comm_unique_id": [_P],
    "rf_comm_init": [_P, _P, _I, _I],
    "rf_comm_allreduce_bucket": [_P, _P, _L, _I, _I, _P],
    "rf_comm_broadcast": [_P, _P, _L, _I, _I, _P],
    "rf_comm_wait": [_P, _P],
    "rf_comm_destroy": [_P],
    "rf_version": [],
}

_lib = None


WGRAD_MAX_GROUP = 48  # RF_WGRAD_MAX_GROUP


class WgradEntry(ctypes.Structure):
    """RfWgradEntry of include/rf_hip.h."""
    _fields_ = [("dy", c_void_p), ("x", c_void_p), ("dw", c_void_p), ("db", c_void_p), ("M", c_int), ("N", c_int),
                ("K", c_int), ("ld_dy", c_int), ("ld_x", c_int), ("splits", c_int), ("kchunk", c_int),
                ("exclusive", c_int), ("dy_bf16", c_int), ("x_bf16", c_int)]


class RowChain(ctypes.Structure):
    """RfRowChain of include/rf_hip.h."""
    _fields_ = ([(n, c_void_p) for n in ("a", "x", "wo", "bo", "g1", "be1", "w1", "b1", "w2", "b2", "g2", "be2", "wp", "bp",
                                         "x1", "y", "proj", "xhat1", "rstd1", "z", "h", "xhat2", "rstd2")]
                + [("d_model", c_int), ("d_ff", c_int), ("n_proj", c_int), ("act", c_int), ("eps", ctypes.c_float),
                   ("drop_site", c_int)])


class RowChainBwd(ctypes.Structure):
    """RfRowChainBwd of include/rf_hip.h."""
    _fields_ = ([(n, c_void_p) for n in ("dproj", "dyin", "wp", "w1", "w2", "g2", "xhat2", "rstd2", "zsrc", "dpre2", "dz", "dg2",
                                         "db2", "wo", "g1", "xhat1", "rstd1", "dpre1", "da", "dg1", "db1")]
                + [("d_model", c_int), ("d_ff", c_int), ("n_proj", c_int), ("act", c_int), ("dx", c_void_p), ("drop_site", c_int),
                   ("pad", c_int)])


class ConvEntry(ctypes.Structure):
    """RfConvEntry of include/rf_hip.h."""
    _fields_ = [("x", c_void_p), ("w_packed", c_void_p), ("bias", c_void_p), ("residual", c_void_p), ("y", c_void_p),
                ("N", c_int), ("H", c_int), ("W", c_int), ("cin", c_int), ("cout", c_int), ("relu", c_int)]


class ConvPairEntry(ctypes.Structure):
    """RfConvPairEntry of include/rf_hip.h."""
    _fields_ = [("x", c_void_p), ("w1_packed", c_void_p), ("bias1", c_void_p), ("w2_packed", c_void_p), ("bias2", c_void_p),
                ("y", c_void_p), ("N", c_int), ("H", c_int), ("W", c_int), ("c", c_int)]


SEQLAYER_MAX_LAYERS, SEQLAYER_MAX_PACK = 8, 64  # RF_SEQLAYER_MAX_LAYERS / RF_SEQLAYER_MAX_PACK


class SeqStack(ctypes.Structure):
    """RfSeqStack of include/rf_hip.h."""
    _fields_ = ([("wpack", c_void_p), ("wpack_stride", c_int64), ("idx", c_void_p * SEQLAYER_MAX_LAYERS),
                 ("idx_stride", c_int64), ("top", c_void_p), ("y", c_void_p)]
                + [(n, c_void_p) for n in ("qkv", "ctx", "xhat1", "rstd1", "x1", "z", "h", "xhat2", "rstd2", "xin")]
                + [("n_layers", c_int), ("flags", c_int)])


class SeqStackBwd(ctypes.Structure):
    """RfSeqStackBwd of include/rf_hip.h."""
    _fields_ = ([("wpack", c_void_p), ("wpack_stride", c_int64)]
                + [(n, c_void_p) for n in ("qkv", "xhat1", "rstd1", "zsrc", "xhat2", "rstd2", "top", "dpre2", "dz", "dpre1",
                                           "dqkv")]
                + [(n, c_void_p * SEQLAYER_MAX_LAYERS) for n in ("dgamma1", "dbeta1", "dgamma2", "dbeta2")]
                + [("n_layers", c_int), ("flags", c_int)])


GATHER_MAX = 8  # RF_GATHER_MAX


FUSE_MAX = 4  # RF_FUSE_MAX


class FuseEntry(ctypes.Structure):
    """RfFuseEntry of include/rf_hip.h."""
    _fields_ = [("base", c_void_p), ("base2", c_void_p), ("src", c_void_p * 3), ("out", c_void_p),
                ("N", c_int), ("Ho", c_int), ("Wo", c_int), ("C", c_int), ("n_src", c_int), ("relu", c_int),
                ("Hi", c_int * 3), ("Wi", c_int * 3)]


class GatherEntry(ctypes.Structure):
    """RfGatherEntry of include/rf_hip.h."""
    _fields_ = [("src", c_void_p), ("dst", c_void_p), ("idx", c_void_p), ("B", c_int), ("T", c_int), ("F", c_int),
                ("pad", c_int), ("frame_bytes", c_int64)]


class SeqPackEntry(ctypes.Structure):
    """RfSeqPackEntry of include/rf_hip.h."""
    _fields_ = [("w", c_void_p), ("out", c_void_p), ("ldw", c_int64), ("N", c_int), ("K", c_int), ("transpose", c_int),
                ("residual", c_int)]


class HipLibraryError(RuntimeError):
    pass


def lib():
    """The loaded library (loads on first use; raises loudly when it is not built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipLibraryError(
                f"{LIB_PATH} is missing: build it with `make -C routeformer_amd/csrc` "
                "(or `python -c 'import __graft_entry__ as g; g.build()'`). "
                "routeformer_amd has no CPU / eager fallback.")
        handle = ctypes.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.argtypes = argtypes
            fn.restype = c_int
        handle.rf_conv3x3_packed_elems.restype = c_int64
        handle.rf_conv3x3s2_packed_elems.restype = c_int64
        handle.rf_pointwise_packed_elems.restype = c_int64
        handle.rf_seqlayer_pack_bytes.restype = c_int64
        handle.rf_seqlayer_bwd_pack_bytes.restype = c_int64
        handle.rf_last_error.restype = ctypes.c_char_p
        handle.rf_last_error.argtypes = []
        _lib = handle
    return _lib


def check(rc: int, what: str):
    if rc != 0:
        msg = lib().rf_last_error()
        raise HipLibraryError(f"{what} failed with code {rc}: {msg.decode() if msg else ''}")


def ptr(t):
    """Device pointer of a tensor (or NULL for None)."""
    return None if t is None else t.data_ptr()
