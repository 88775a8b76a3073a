"""ctypes binding of libdzn_hip.so (the C ABI declared in include/dzn.h, include/dzn_ops.h).

There is NO CPU fallback: if the shared object is missing, or a call is made without a
HIP device, this module raises.  The oracle under oracle/ is test infrastructure and is
never imported from here.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

DZN_MAX_CONV = 8
DZN_MAX_LAYERS = 32
DZN_MAX_HEADS = 16

DZN_PREC_F32 = 0
DZN_PREC_BF16 = 1
DZN_PREC_F32_SPLIT = 2
DZN_PREC_F32_H2 = 3
DZN_PREC_F16 = 4

DZN_ACT_NONE, DZN_ACT_GELU, DZN_ACT_SWISH, DZN_ACT_RELU = 0, 1, 2, 3

DZN_F32, DZN_F64, DZN_I64 = 0, 1, 2

ERRORS = {0: "ok", -1: "invalid argument", -2: "out of memory", -3: "HIP runtime error",
          -4: "bad call order", -5: "missing state_dict key"}


class DznError(RuntimeError):
    pass


class DznConfig(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32),
        ("precision", C.c_int32),
        ("max_batch", C.c_int32),
        ("max_samples", C.c_int32),
        ("extractor_layer_norm", C.c_int32),
        ("normalize_waveform", C.c_int32),
        ("n_conv", C.c_int32),
        ("conv_ch", C.c_int32 * DZN_MAX_CONV),
        ("conv_k", C.c_int32 * DZN_MAX_CONV),
        ("conv_s", C.c_int32 * DZN_MAX_CONV),
        ("embed_dim", C.c_int32),
        ("total_heads", C.c_int32),
        ("n_layers", C.c_int32),
        ("layer_norm_first", C.c_int32),
        ("pos_conv_kernel", C.c_int32),
        ("pos_conv_groups", C.c_int32),
        ("num_buckets", C.c_int32),
        ("max_distance", C.c_int32),
        ("use_attention", C.c_int32 * DZN_MAX_LAYERS),
        ("n_heads", C.c_int32 * DZN_MAX_LAYERS),
        ("head_idx", (C.c_int32 * DZN_MAX_HEADS) * DZN_MAX_LAYERS),
        ("use_ffn", C.c_int32 * DZN_MAX_LAYERS),
        ("ffn_dim", C.c_int32 * DZN_MAX_LAYERS),
        ("attention_in", C.c_int32),
        ("ffn_hidden", C.c_int32),
        ("conf_heads", C.c_int32),
        ("conf_layers", C.c_int32),
        ("conf_kernel", C.c_int32),
        ("n_classes", C.c_int32),
        ("max_speakers_per_chunk", C.c_int32),
        ("max_speakers_per_frame", C.c_int32),
        ("has_embedding", C.c_int32),
        ("embed_out_dim", C.c_int32),
        ("num_mel_bins", C.c_int32),
        ("reserved", C.c_int32 * 16),
    ]


class DznGemmDesc(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("W", C.c_void_p), ("W16", C.c_void_p), ("C", C.c_void_p),
        ("bias", C.c_void_p), ("R", C.c_void_p), ("WS", C.c_void_p),
        ("a_rowoff", C.c_void_p), ("c_rowoff", C.c_void_p),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
        ("lda", C.c_int64), ("kc", C.c_int32), ("ldk", C.c_int64), ("ldw", C.c_int32),
        ("ldc", C.c_int64), ("ldws", C.c_int64),
        ("act", C.c_int32), ("alpha", C.c_float), ("post_relu", C.c_int32),
        ("ws_w", C.c_float), ("ws_init", C.c_int32),
        ("nz", C.c_int32), ("zdiv", C.c_int32),
        ("a_z0", C.c_int64), ("a_z1", C.c_int64), ("w_z0", C.c_int64), ("w_z1", C.c_int64),
        ("c_z0", C.c_int64), ("c_z1", C.c_int64), ("b_z0", C.c_int64), ("b_z1", C.c_int64),
        ("precision", C.c_int32),
        ("alg_flops", C.c_double),
        ("a_bf16", C.c_int32), ("c_bf16", C.c_int32), ("r_bf16", C.c_int32),
        ("W3", C.c_void_p),
        ("a_split3", C.c_int32), ("a_plane", C.c_int64),
        ("ln_stats", C.c_void_p), ("ln_colsum", C.c_void_p),
        ("W2h", C.c_void_p), ("col_scale", C.c_void_p), ("a_amax", C.c_void_p), ("c_amax", C.c_void_p),
        ("amax_unit", C.c_int32),
        ("stat_partial", C.c_void_p), ("stat_final", C.c_void_p), ("stat_C", C.c_int32), ("stat_eps", C.c_float),
        ("z_count", C.c_void_p), ("z_list", C.c_void_p), ("ln_centered", C.c_int32),
        ("Wmx", C.c_void_p), ("col_scale_mx", C.c_void_p),
        ("amax_count", C.c_int32),
        ("kv_planes", C.c_void_p), ("kv_plane_stride", C.c_int64), ("kv_scale", C.c_void_p), ("kv_ld", C.c_int32),
        ("kv_col0", C.c_int32),
    ]


class DznProfEntry(C.Structure):
    _fields_ = [("name", C.c_char * 64), ("launches", C.c_int64), ("ms", C.c_double),
                ("flops", C.c_double), ("bytes", C.c_double)]


def lib_path() -> Path:
    env = os.environ.get("DZN_HIP_LIB")
    if env:
        return Path(env)
    return Path(__file__).resolve().parent / "lib" / "libdzn_hip.so"


_LIB = None


def load() -> C.CDLL:
    """Load libdzn_hip.so.  Raises DznError (never falls back) if it is not built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    p = lib_path()
    if not p.exists():
        raise DznError(
            f"{p} not found: build the HIP extension first (python -m diarizen_amd.build). "
            "diarizen_amd has no CPU fallback.")
    try:
        import torch  # noqa: F401  -- load torch's HIP runtime first so it is shared
    except Exception:  # pragma: no cover
        pass
    lib = C.CDLL(str(p))
    vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float

    def sig(name, res, args):
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args

    sig("dzn_create", i32, [C.POINTER(DznConfig), C.POINTER(vp)])
    sig("dzn_load_tensor", i32, [vp, C.c_char_p, vp, C.POINTER(i64), i32, i32])
    sig("dzn_finalize_weights", i32, [vp])
    sig("dzn_num_frames", i32, [vp, i32])
    sig("dzn_segment_forward", i32, [vp, vp, i32, i32, vp, vp, vp])
    sig("dzn_embed_forward", i32, [vp, vp, vp, i32, i32, i32, i32, vp, vp])
    sig("dzn_prepare_masks", i32, [vp, vp, i32, i32, i32, i32, i32, vp, vp, vp])
    sig("dzn_speaker_count", i32, [vp, i32, i32, i32, vp, i32, vp, vp, vp])
    sig("dzn_cluster_activations", i32, [vp, vp, i32, i32, i32, vp, i32, i32, vp, vp])
    sig("dzn_debug_fetch", i32, [vp, C.c_char_p, vp, i64, C.POINTER(i64)])
    sig("dzn_embed_skip_stats", i32, [vp, C.POINTER(i64), C.POINTER(i64)])
    sig("dzn_num_ignored", i32, [vp])
    sig("dzn_workspace_bytes", i64, [vp])
    sig("dzn_last_error", C.c_char_p, [vp])
    sig("dzn_destroy", i32, [vp])
    sig("dzn_version", C.c_char_p, [])
    sig("dzn_profile_enable", i32, [i32])
    sig("dzn_profile_collect", i32, [C.POINTER(DznProfEntry), i32, C.POINTER(i32)])
    sig("dzn_profile_reserve", i32, [i32])
    sig("dzn_op_relpos_bucket", i32, [i32, i32, i32])
    sig("dzn_linkage_centroid", i32, [vp, i32, i32, vp, i32])
    sig("dzn_cdist_cosine", i32, [vp, i32, i32, vp, i32, vp, i32])
    sig("dzn_host_workspace_release", i32, [i32])
    sig("dzn_host_workspace_bytes", i64, [i32])
    sig("dzn_vbx_create", i32, [vp, vp, vp, i32, i32, i32, i32, C.POINTER(C.c_void_p)])
    sig("dzn_vbx_stats", i32, [vp, vp])
    sig("dzn_vbx_estep", i32, [vp, vp, vp, vp, C.c_double, C.POINTER(C.c_double)])
    sig("dzn_vbx_gamma", i32, [vp, vp])
    sig("dzn_vbx_destroy", i32, [vp])
    sig("dzn_op_gemm", i32, [C.POINTER(DznGemmDesc), vp])
    sig("dzn_op_split_weights", i32, [vp, i64, i32, i64, vp, vp])
    sig("dzn_op_split_weights_h2", i32, [vp, i64, i32, i64, vp, vp, vp])
    sig("dzn_op_split_weights_mx", i32, [vp, i64, i32, i64, vp, vp, vp])
    sig("dzn_op_set_gemm_mx_cfg", i32, [C.c_char_p])
    sig("dzn_checked_status", i32, [vp, i32])
    sig("dzn_flac_info", i32, [vp, C.c_size_t, vp, vp, vp, vp, vp])
    sig("dzn_flac_decode", i32, [vp, C.c_size_t, vp, i64, vp])
    sig("dzn_op_set_gemm_cfg", i32, [C.c_char_p])
    sig("dzn_op_amax", i32, [vp, i64, vp, vp])
    sig("dzn_op_split_rows", i32, [vp, vp, i64, i64, i32, vp])
    sig("dzn_op_conv3x3_c32", i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp])
    sig("dzn_op_resblock_ws", i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, f32, f32, i32, i32, i32, i32, vp])
    sig("dzn_op_resblock32_fused", i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, f32, f32, i32, i32, i32, vp])
    sig("dzn_op_set_resblock_np", i32, [i32])
    sig("dzn_op_set_attention_noskip", i32, [i32])
    sig("dzn_op_conv3x3_c32_h2", i32, [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp])
    sig("dzn_op_layernorm", i32, [vp, i64, vp, i64, vp, vp, i64, i32, i32, f32, i32, vp])
    sig("dzn_op_gate", i32, [vp, i64, vp, vp, vp, vp, i64, i32, vp])
    sig("dzn_op_row_stats", i32, [vp, i64, i64, i32, f32, vp, vp])
    sig("dzn_op_gate_stats", i32, [vp, i64, vp, vp, vp, vp, vp, vp, vp, i64, i32, f32, vp])
    sig("dzn_op_attention_h2", i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, f32, vp, vp])
    sig("dzn_op_set_attention_qb", i32, [i32])
    sig("dzn_op_set_attention_prefetch", i32, [i32])
    sig("dzn_op_attention_planes", i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, f32, vp, vp, vp, vp])
    sig("dzn_op_attention", i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, f32, i32, vp])
    _LIB = lib
    return lib


EXPORTED = [
    "dzn_create", "dzn_load_tensor", "dzn_finalize_weights", "dzn_num_frames",
    "dzn_segment_forward", "dzn_embed_forward", "dzn_prepare_masks", "dzn_speaker_count", "dzn_cluster_activations", "dzn_debug_fetch", "dzn_embed_skip_stats", "dzn_num_ignored",
    "dzn_workspace_bytes", "dzn_last_error", "dzn_destroy", "dzn_version", "dzn_linkage_centroid", "dzn_cdist_cosine", "dzn_host_workspace_release",
    "dzn_host_workspace_bytes",
    "dzn_flac_info", "dzn_flac_decode", "dzn_vbx_create", "dzn_vbx_stats", "dzn_vbx_estep", "dzn_vbx_gamma", "dzn_vbx_destroy",
    "dzn_op_gemm", "dzn_op_split_weights", "dzn_op_split_weights_h2", "dzn_op_split_weights_mx", "dzn_op_set_gemm_mx_cfg", "dzn_checked_status", "dzn_op_set_gemm_cfg", "dzn_op_amax", "dzn_op_conv3x3_c32", "dzn_op_conv3x3_c32_h2", "dzn_op_resblock32_fused", "dzn_op_resblock_ws", "dzn_op_set_resblock_np", "dzn_op_set_attention_noskip", "dzn_op_split_rows", "dzn_op_layernorm", "dzn_op_row_stats", "dzn_op_gate", "dzn_op_gate_stats", "dzn_op_attention", "dzn_op_attention_h2", "dzn_op_attention_planes", "dzn_op_set_attention_qb", "dzn_op_set_attention_prefetch",
    "dzn_profile_enable", "dzn_profile_collect", "dzn_profile_reserve", "dzn_op_relpos_bucket",
]


def profile_enable(on: bool) -> None:
    check(load().dzn_profile_enable(int(on)), None, "dzn_profile_enable")


def profile_reserve(n_events: int) -> None:
    """pre-create HIP events so that a timed region with the profiler on never allocates one"""
    check(load().dzn_profile_reserve(int(n_events)), None, "dzn_profile_reserve")


def profile_collect() -> list:
    """[{name, launches, ms, flops, bytes}] aggregated per kernel class since the last collect."""
    lib = load()
    arr = (DznProfEntry * 512)()
    n = C.c_int32(0)
    check(lib.dzn_profile_collect(arr, 512, C.byref(n)), None, "dzn_profile_collect")
    return [dict(name=arr[i].name.decode(), launches=arr[i].launches, ms=arr[i].ms,
                 flops=arr[i].flops, bytes=arr[i].bytes) for i in range(min(n.value, 512))]


def check(rc: int, handle=None, what: str = "") -> None:
    if rc == 0:
        return
    msg = ERRORS.get(rc, f"error {rc}")
    try:
        detail = load().dzn_last_error(handle)
        if detail:
            msg += ": " + detail.decode(errors="replace")
    except Exception:  # pragma: no cover
        pass
    if rc == -2:
        # the reference turns device OOM into MemoryError (PA/core/inference.py:216-221)
        raise MemoryError(f"{what}: {msg}")
    raise DznError(f"{what}: {msg}")
