"""ctypes binding of libudt_kernels.so (C ABI declared in include/udt_kernels.h).

The library is built in-tree by ``udifftext_amd.build``.  There is NO fallback: if the shared object is
missing, or a compute entry point is called without a gfx950 device, the call raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libudt_kernels.so")

# flags (mirror include/udt_kernels.h)
GEMM_OUT_F32 = 1 << 0
DTYPE_F32, DTYPE_BF16, DTYPE_FP8_E4M3 = 0, 1, 2
GEMM_GEGLU = 1 << 1
GEMM_RELU = 1 << 2
GEMM_TRANSPOSED = 1 << 3
GEMM_CONV = 1 << 4
GEMM_SILU_OUT = 1 << 5
GEMM_MX8 = 1 << 7

PROF_CONV3X3, PROF_GEMM, PROF_ATTN, PROF_XATTN, PROF_NORM, PROF_ELEMENTWISE = range(6)


class GemmDesc(C.Structure):
    _fields_ = [
        ("a", C.c_void_p), ("a2", C.c_void_p), ("w", C.c_void_p), ("bias", C.c_void_p),
        ("residual", C.c_void_p), ("rowvec", C.c_void_p), ("out", C.c_void_p),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
        ("lda", C.c_int32), ("ldo", C.c_int32), ("ldr", C.c_int32), ("ldw", C.c_int32),
        ("batch", C.c_int32),
        ("stride_a", C.c_int64), ("stride_w", C.c_int64), ("stride_out", C.c_int64), ("stride_res", C.c_int64),
        ("Hin", C.c_int32), ("Win", C.c_int32), ("C1", C.c_int32), ("C2", C.c_int32),
        ("Hout", C.c_int32), ("Wout", C.c_int32),
        ("ksize", C.c_int32), ("stride", C.c_int32), ("pad_t", C.c_int32), ("pad_l", C.c_int32),
        ("upsample", C.c_int32), ("rows_per_batch", C.c_int32), ("ld_rowvec", C.c_int32), ("flags", C.c_int32),
        ("alpha", C.c_float), ("colscale", C.c_void_p), ("in_scsh", C.c_void_p), ("in_act", C.c_int32), ("colstats", C.c_void_p),
        ("cu_share", C.c_int32), ("ln_colsum", C.c_void_p), ("ln_eps", C.c_float),
        ("a_scale", C.c_void_p), ("q8_out", C.c_void_p), ("q8_scale", C.c_void_p), ("ld_q8", C.c_int32),
        ("q8_fixed_col", C.c_int32), ("q8_fixed_mul", C.c_float),
        ("rowstat_out", C.c_void_p), ("rowstat_in", C.c_void_p), ("rowstat_in_parts", C.c_int32),
    ]


# every symbol include/udt_kernels.h declares: name -> (restype, argtypes)
_vp, _i32, _i64, _f32, _fp = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_void_p
SYMBOLS = {
    "udt_gemm_workspace_bytes": (C.c_size_t, [C.POINTER(GemmDesc)]),
    "udt_gemm": (C.c_int, [C.POINTER(GemmDesc), _vp, C.c_size_t, _vp]),
    "udt_gemm_fwd": (C.c_int, [C.POINTER(GemmDesc), _vp, C.c_size_t, _vp]),
    "udt_conv1x1_fwd": (C.c_int, [C.POINTER(GemmDesc), _vp, C.c_size_t, _vp]),
    "udt_workspace_bytes": (C.c_size_t, [C.POINTER(GemmDesc)]),
    "udt_pack_linear": (C.c_int, [_fp, _fp, _i32, _i32, _i32, _i32, C.POINTER(_vp), _vp]),
    "udt_pack_conv": (C.c_int, [_fp, _fp, _i32, _i32, _i32, _i32, C.POINTER(_i32), _i32, _i32, C.POINTER(_vp), _vp]),
    "udt_packed_weight": (_vp, [_vp]),
    "udt_packed_bias": (_vp, [_vp]),
    "udt_packed_colscale": (_vp, [_vp]),
    "udt_packed_dim": (_i32, [_vp, _i32]),
    "udt_free_packed": (C.c_int, [_vp]),
    "udt_check_async_error": (C.c_int, [_vp, C.c_size_t, _vp]),
    "udt_gemm_colstats_rows": (_i32, [C.POINTER(GemmDesc)]),
    "udt_gemm_colstats_slots": (_i32, [C.POINTER(GemmDesc)]),
    "udt_gemm_rowstat_parts": (_i32, [C.POINTER(GemmDesc)]),
    "udt_gemm_q8_ok": (_i32, [C.POINTER(GemmDesc)]),
    "udt_gemm_in_scsh_ok": (_i32, [C.POINTER(GemmDesc)]),
    "udt_gn_silu_conv3x3_fwd": (C.c_int, [C.POINTER(GemmDesc), _vp, C.c_size_t, _vp]),
    "udt_ln_gemm_fwd": (C.c_int, [C.POINTER(GemmDesc), _vp, C.c_size_t, _vp]),
    "udt_gn_finalize": (C.c_int, [_fp, _i32, _i32, _fp, _i32, _i32, _fp, _fp, _fp, _i32, _i64, _i32, _f32, _vp]),
    "udt_gn_apply_scsh": (C.c_int, [_vp, _vp, _vp, _fp, _i32, _i64, _i32, _i32, _i32, _vp]),
    "udt_gn_strip_stats": (C.c_int, [_vp, _vp, _vp, _fp, _i32, _fp, _i32, _fp, _fp, _i32, _i64, _i32, _i32, _i32, _f32, _i32, _vp]),
    "udt_attn_fwd": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32,
                               _i64, _i64, _i64, _i64, _f32, _vp]),
    "udt_attn_rowv_fwd": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32,
                                    _i64, _i64, _i64, _i64, _f32, _vp]),
    "udt_attn_rowv_q8_fwd": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32,
                                       _i64, _i64, _i64, _i64, _f32, _vp, _vp, _i32, _vp]),
    "udt_attn_mx8_fwd": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _f32, _vp, _vp, _i32, _vp]),
    "udt_attn512_fwd": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i64, _i64, _i64, _i64, _f32, _vp]),
    "udt_attn512_workspace_bytes": (C.c_size_t, [_i32, _i32, _i32]),
    "udt_attn512_split_fwd": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i64, _i64, _i64, _i64, _f32,
                                        _vp, C.c_size_t, _vp]),
    "udt_xattn_fwd": (C.c_int, [_vp, _vp, _vp, _vp, _fp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _vp]),
    "udt_mattn_fwd": (C.c_int, [_vp, _vp, _vp, _vp, _fp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32,
                                _i64, _i64, _i64, _i64, _f32, _vp]),
    "udt_softmax_rows": (C.c_int, [_vp, _i64, _i32, _i32, _vp]),
    "udt_tattn_hp": (_i32, [_i32]),
    "udt_tattn_prepare": (C.c_int, [_vp, _i32, _vp, _i32, _vp, _i32, _fp, _fp, _vp, _fp, _vp, _i32, _i32, _i32, _i32, _f32, _vp]),
    "udt_tattn_fused": (C.c_int, [_vp, _vp, _vp, _fp, _vp, _fp, _i32, _i32, _i32, _i32, _i32, _f32, _vp]),
    "udt_tattn_rowstat_parts": (_i32, [_i32, _i32, _i32]),
    "udt_tattn_fused_q8": (C.c_int, [_vp, _vp, _vp, _fp, _vp, _fp, _i32, _i32, _i32, _i32, _i32, _f32, _vp, _vp, _fp, _vp]),
    "udt_gn_nchunks": (_i32, [_i64, _i32]),
    "udt_gn_stats": (C.c_int, [_vp, _vp, _fp, _i32, _i64, _i32, _i32, _i32, _vp]),
    "udt_gn_apply": (C.c_int, [_vp, _vp, _vp, _fp, _fp, _fp, _i32, _i64, _i32, _i32, _i32, _f32, _i32, _vp]),
    "udt_gn_strip_ok": (_i32, [_i32, _i64, _i32, _i32, _i32]),
    "udt_gn_strip": (C.c_int, [_vp, _vp, _vp, _fp, _fp, _i32, _i64, _i32, _i32, _i32, _f32, _i32, _vp]),
    "udt_layernorm": (C.c_int, [_vp, _vp, _fp, _fp, _i64, _i32, _f32, _vp]),
    "udt_unet_input": (C.c_int, [_fp, _vp, _i32, _i32, _i32, _f32, _vp]),
    "udt_cfg_euler_step": (C.c_int, [_fp, _fp, _fp, _i32, _i32, _i32, _f32, _f32, _f32, _f32, _vp]),
    "udt_sampler_step": (C.c_int, [_fp, _fp, _fp, _i32, _i32, _i32, _f32, _f32, _f32, _f32, _vp]),
    "udt_posterior_sample": (C.c_int, [_fp, _fp, _fp, _i32, _i32, _i32, _f32, _vp]),
    "udt_nchw_to_nhwc": (C.c_int, [_fp, _vp, _i32, _i32, _i64, _i32, _f32, _vp]),
    "udt_nhwc_to_nchw": (C.c_int, [_vp, _fp, _i32, _i32, _i64, _i32, _i32, _vp]),
    "udt_nhwc_set_channels": (C.c_int, [_fp, _vp, _i32, _i32, _i64, _i32, _i32, _vp]),
    "udt_embed_tokens": (C.c_int, [_vp, _fp, _fp, _vp, _i32, _i32, _i32, _vp]),
    "udt_timestep_embedding": (C.c_int, [_fp, _vp, _i32, _i32, _vp]),
    "udt_mask_downsample": (C.c_int, [_fp, _fp, _i32, _i32, _i32, _vp]),
    "udt_local_loss": (C.c_int, [_fp, _fp, _fp, _fp, _fp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "udt_local_loss_tiled": (C.c_int, [_fp, _fp, _fp, _fp, _fp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "udt_attn_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _fp, _fp, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _vp]),
    "udt_xattn_bwd": (C.c_int, [_vp, _vp, _fp, _fp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _vp]),
    "udt_local_loss_bwd": (C.c_int, [_fp, _fp, _fp, _fp, _fp, _fp, _fp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _vp]),
    "udt_layernorm_bwd": (C.c_int, [_vp, _vp, _fp, _vp, _vp, _i64, _i32, _f32, _vp]),
    "udt_gn_bwd": (C.c_int, [_vp, _vp, _fp, _fp, _vp, _vp, _fp, _i32, _i32, _i32, _i32, _f32, _i32, _vp]),
    "udt_geglu_fwd": (C.c_int, [_vp, _vp, _i64, _i32, _vp]),
    "udt_geglu_bwd": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _vp]),
    "udt_sum2x2_bf16": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "udt_axpy_f32": (C.c_int, [_fp, _fp, _f32, _i64, _vp]),
    "udt_center_tokens": (C.c_int, [_fp, _vp, _i32, _i32, _i32, _vp]),
    "udt_transpose_bf16": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "udt_reduce_rows_f32": (C.c_int, [_fp, _fp, _i32, _i64, _i32, _vp]),
    "udt_colparts": (_i32, [_i64]),
    "udt_colsum_bf16": (C.c_int, [_vp, _fp, _fp, _i64, _i32, _vp]),
    "udt_ln_param_grad": (C.c_int, [_vp, _vp, _fp, _fp, _i64, _i32, _f32, _vp]),
    "udt_xattn_bwd_kv": (C.c_int, [_vp, _vp, _fp, _fp, _vp, _vp, _vp, _fp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _vp]),
    "udt_xattn_kv_splits": (_i32, [_i32]),
    "udt_wgrad_splits": (_i32, [_i64, _i32, _i32]),
    "udt_wgrad_bf16": (C.c_int, [_vp, _vp, _fp, _fp, _i64, _i32, _i32, _i32, _i32, _vp]),
    "udt_local_loss_seg_bwd": (C.c_int, [_fp, _fp, _fp, _fp, _fp, _fp, _fp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _vp]),
    "udt_diff_loss_grad": (C.c_int, [_fp, _fp, _fp, _fp, _vp, _fp, _i32, _i32, _i32, _i32, _vp]),
    "udt_adamw_f32": (C.c_int, [_fp, _fp, _fp, _fp, _i64, _f32, _f32, _f32, _f32, _f32, _i32, _f32, _vp]),
    "udt_add_bf16": (C.c_int, [_vp, _vp, _i64, _vp]),
    "udt_debug_set": (C.c_int, [C.c_char_p, _i32]),
    "udt_bias_add_bf16": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _vp]),
    "udt_version": (C.c_char_p, []),
    "udt_status_string": (C.c_char_p, [C.c_int]),
    "udt_last_hip_error": (C.c_int, []),
    "udt_device_arch_ok": (C.c_int, []),
    "udt_prof_enable": (C.c_int, [C.c_uint32]),
    "udt_prof_reset": (C.c_int, []),
    "udt_prof_trace": (C.c_int, [_i32]),
    "udt_prof_dump": (C.c_int, [C.c_char_p]),
    "udt_prof_get": (C.c_int, [_i32, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
}

_lib = None


class UdtError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load the shared object (once) and type every entry point.  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise UdtError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `python -m udifftext_amd.build` "
            "(or __graft_entry__.build()). There is no CPU fallback for the product path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)   # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status: int, what: str = "") -> None:
    if status == 0:
        return
    lib = load()
    msg = lib.udt_status_string(status).decode()
    if status == -4:
        msg += f" [hipError {lib.udt_last_hip_error()}]"
    if status in (-1, -2):
        raise ValueError(f"{what}: {msg}")
    raise UdtError(f"{what}: {msg}")
