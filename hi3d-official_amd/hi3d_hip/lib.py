"""ctypes binding of the C ABI declared in include/hi3d_hip.h.

The product path has exactly one backend: this library.  If it cannot be loaded the
import of `ops` fails loudly -- there is no PyTorch / CPU fallback.
"""
import ctypes as C
import os

# torch must be loaded first: it ships its own libamdhip64 and our library has to bind to
# that same HIP runtime instance (device pointers and streams come from torch).
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libhi3d_hip.so")

EXPORTS = [
    "hi3d_abi_version", "hi3d_last_error", "hi3d_gemm_bf16", "hi3d_gemm_set_workspace", "hi3d_gemm_set_workspace_for_stream", "hi3d_gemm_reload_env", "hi3d_debug_gemm_launch_info", "hi3d_debug_gemm_launch_info_on", "hi3d_attn_d64", "hi3d_attn_d64_v", "hi3d_attn_d512",
    "hi3d_transpose_v", "hi3d_attn_temporal_d64", "hi3d_attn_fp8_workspace_bytes", "hi3d_attn_quant_qk", "hi3d_attn_d64_fp8qk", "hi3d_attn_fp8_v_workspace_bytes", "hi3d_attn_quant_v", "hi3d_attn_d64_fp8", "hi3d_gn_partial_blocks",
    "hi3d_gn_workspace_floats", "hi3d_groupnorm_silu", "hi3d_groupnorm_silu_cat2", "hi3d_groupnorm_silu_from_partials", "hi3d_gemm_gn_partial_supported", "hi3d_gemm_last_gn_fused", "hi3d_groupnorm_partial_sums", "hi3d_groupnorm_apply_sums", "hi3d_layernorm",
    "hi3d_concat_channels", "hi3d_timestep_embedding", "hi3d_silu_f32_to_bf16",
    "hi3d_cfg_prepare", "hi3d_sampler_step", "hi3d_cfg_update_x", "hi3d_sampler_step_dev", "hi3d_nchw_f32_to_nhwc_bf16",
    "hi3d_nhwc_to_nchw_f32", "hi3d_vae_latent_prepare", "hi3d_softmax_rows", "hi3d_vae_posterior", "hi3d_v02_blend", "hi3d_time_mix_small", "hi3d_time_mix_small_k3", "hi3d_ffn_geglu", "hi3d_ffn_geglu_ln", "hi3d_groupnorm_fold_linear", "hi3d_groupnorm_fold_linear_from_partials",
    "hi3d_act_bf16", "hi3d_l2_normalize_rows", "hi3d_add_act_bf16", "hi3d_dpt_stem_conv", "hi3d_pool2_nhwc",
    "hi3d_resize_bilinear_nhwc", "hi3d_dpt_head_out", "hi3d_depth_normalize_unshuffle", "hi3d_resample_axis", "hi3d_permute_rows",
]

ABI_VERSION = 3          # == HI3D_ABI_VERSION of the include/hi3d_hip.h this binding was written against (GemmDesc below is that layout)
A_DENSE, A_CONV3X3, A_CONVT3 = 0, 1, 2
EPI_AFFINE, EPI_GEGLU = 0, 1


class GemmDesc(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("W", C.c_void_p), ("bias", C.c_void_p), ("rowvec", C.c_void_p),
        ("R1", C.c_void_p), ("R2", C.c_void_p), ("a1", C.c_void_p), ("a2", C.c_void_p),
        ("out", C.c_void_p),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
        ("lda", C.c_int32), ("ldo", C.c_int32), ("ldr1", C.c_int32), ("ldr2", C.c_int32),
        ("ldrv", C.c_int32), ("ldw", C.c_int32), ("rows_per_group", C.c_int32),
        ("amode", C.c_int32), ("epi", C.c_int32), ("out_fp32", C.c_int32),
        ("Hin", C.c_int32), ("Win", C.c_int32), ("Cin", C.c_int32), ("Hout", C.c_int32),
        ("Wout", C.c_int32), ("stride", C.c_int32), ("up2x", C.c_int32),
        ("T", C.c_int32), ("HW", C.c_int32), ("tile_n", C.c_int32), ("pad_br_only", C.c_int32),
        ("A2", C.c_void_p), ("K1", C.c_int32), ("lda2", C.c_int32), ("gn_partial", C.c_void_p), ("w_group_stride", C.c_int64),
        ("conv_ntap", C.c_int32), ("conv_taps", C.c_uint32), ("conv_phase", C.c_int32),
    ]


class Hi3dError(RuntimeError):
    pass


_lib = None


def load():
    """Load libhi3d_hip.so (once). Raises if it is missing -- never falls back."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise Hi3dError(
            f"{LIB_PATH} not found: build it with `python hi3d-official_amd/build.py` "
            "(or __graft_entry__.build()). There is no fallback backend.")
    lib = C.CDLL(LIB_PATH)
    i32, i64, f32, vp = C.c_int32, C.c_int64, C.c_float, C.c_void_p
    sig = {
        "hi3d_abi_version": (C.c_int, []),
        "hi3d_last_error": (C.c_char_p, []),
        "hi3d_gemm_bf16": (C.c_int, [C.POINTER(GemmDesc), vp]),
        "hi3d_gemm_set_workspace": (C.c_int, [vp, i64]),
        "hi3d_gemm_set_workspace_for_stream": (C.c_int, [vp, i64, vp]),
        "hi3d_gemm_reload_env": (C.c_int, []),
        "hi3d_debug_gemm_launch_info": (C.c_int, [C.POINTER(GemmDesc), vp, C.POINTER(C.c_int32)]),
        "hi3d_debug_gemm_launch_info_on": (C.c_int, [C.POINTER(GemmDesc), vp, vp, C.POINTER(C.c_int32)]),
        "hi3d_attn_d64": (C.c_int, [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, f32, vp]),
        "hi3d_attn_d64_v": (C.c_int, [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, f32, vp]),
        "hi3d_attn_d512": (C.c_int, [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, f32, vp]),
        "hi3d_attn_fp8_workspace_bytes": (i64, [i32, i32, i32]),
        "hi3d_attn_quant_qk": (C.c_int, [vp, vp, i32, i32, i32, i32, vp]),
        "hi3d_attn_d64_fp8qk": (C.c_int, [vp, vp, vp, i32, i32, i32, i32, i32, vp]),
        "hi3d_attn_fp8_v_workspace_bytes": (i64, [i32, i32, i32]),
        "hi3d_attn_quant_v": (C.c_int, [vp, vp, i32, i32, i32, i32, vp]),
        "hi3d_attn_d64_fp8": (C.c_int, [vp, vp, vp, i32, i32, i32, i32, vp]),
        "hi3d_transpose_v": (C.c_int, [vp, vp, i32, i32, i32, i32, i32, vp]),
        "hi3d_attn_temporal_d64": (C.c_int, [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, f32, vp]),
        "hi3d_gn_partial_blocks": (i32, [i32, i32]),
        "hi3d_gn_workspace_floats": (i64, [i32, i32, i32]),
        "hi3d_groupnorm_silu": (C.c_int, [vp, vp, vp, vp, vp, i32, i32, i32, f32, i32, vp]),
        "hi3d_groupnorm_silu_from_partials": (C.c_int, [vp, vp, vp, vp, vp, i32, i32, i32, f32, i32, vp]),
        "hi3d_gemm_gn_partial_supported": (C.c_int, [C.POINTER(GemmDesc), vp]),
        "hi3d_gemm_last_gn_fused": (C.c_int, []),
        "hi3d_groupnorm_silu_cat2": (C.c_int, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, i32, vp]),
        "hi3d_groupnorm_partial_sums": (C.c_int, [vp, vp, vp, i32, i32, i32, vp]),
        "hi3d_groupnorm_apply_sums": (C.c_int, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i64, f32, i32, vp]),
        "hi3d_layernorm": (C.c_int, [vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, vp]),
        "hi3d_concat_channels": (C.c_int, [vp, vp, vp, i64, i32, i32, vp]),
        "hi3d_timestep_embedding": (C.c_int, [vp, vp, i32, i32, f32, i32, vp]),
        "hi3d_silu_f32_to_bf16": (C.c_int, [vp, vp, i64, vp]),
        "hi3d_cfg_prepare": (C.c_int, [vp, vp, vp, vp, i32, i32, i32, i32, f32, vp]),
        "hi3d_sampler_step": (C.c_int, [vp, vp, vp, i32, i32, i32, f32, f32, vp]),
        "hi3d_cfg_update_x": (C.c_int, [vp, vp, vp, vp, i32, i32, i32, vp]),
        "hi3d_sampler_step_dev": (C.c_int, [vp, vp, vp, vp, vp, i32, i32, i32, vp]),
        "hi3d_nchw_f32_to_nhwc_bf16": (C.c_int, [vp, vp, i32, i32, i32, i32, vp]),
        "hi3d_nhwc_to_nchw_f32": (C.c_int, [vp, vp, i32, i32, i32, i32, i32, vp]),
        "hi3d_vae_latent_prepare": (C.c_int, [vp, vp, vp, vp, i32, i32, i32, i32, vp]),
        "hi3d_softmax_rows": (C.c_int, [vp, vp, i32, i32, i32, i32, f32, vp]),
        "hi3d_vae_posterior": (C.c_int, [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
        "hi3d_v02_blend": (C.c_int, [vp, vp, vp, i64, f32, f32, vp]),
        "hi3d_time_mix_small": (C.c_int, [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
        "hi3d_time_mix_small_k3": (C.c_int, [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
        "hi3d_ffn_geglu": (C.c_int, [vp] * 10 + [i32] * 7 + [vp]),
        "hi3d_groupnorm_fold_linear": (C.c_int, [vp] * 4 + [C.c_float] + [i32] * 3 + [vp, i32, vp, i32, vp, vp, vp]),
        "hi3d_groupnorm_fold_linear_from_partials": (C.c_int, [vp] * 3 + [C.c_float] + [i32] * 3 + [vp, i32, vp, i32, vp, vp, vp]),
        "hi3d_ffn_geglu_ln": (C.c_int, [vp] * 3 + [C.c_float, vp, i32] + [vp] * 9 + [i32] * 7 + [vp]),
        "hi3d_act_bf16": (C.c_int, [vp, i64, i32, vp]),
        "hi3d_l2_normalize_rows": (C.c_int, [vp, i32, i32, vp]),
        "hi3d_add_act_bf16": (C.c_int, [vp, vp, vp, i64, i32, vp]),
        "hi3d_dpt_stem_conv": (C.c_int, [vp, vp, vp, i32, i32, i32, vp]),
        "hi3d_pool2_nhwc": (C.c_int, [vp, vp, i32, i32, i32, i32, i32, vp]),
        "hi3d_resize_bilinear_nhwc": (C.c_int, [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
        "hi3d_dpt_head_out": (C.c_int, [vp, vp, C.c_float, vp, i64, i32, vp]),
        "hi3d_depth_normalize_unshuffle": (C.c_int, [vp, vp, i32, i32, i32, i32, vp]),
        "hi3d_permute_rows": (C.c_int, [vp, vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), i32, vp]),
        "hi3d_resample_axis": (C.c_int, [vp, vp, vp, vp, i32, i64, i32, i32, i32, vp, vp, i32, i32, vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    if lib.hi3d_abi_version() != ABI_VERSION:
        raise Hi3dError(f"ABI version mismatch: library reports {lib.hi3d_abi_version()}, this binding is written for {ABI_VERSION}")
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().hi3d_last_error().decode(errors="replace")
        raise Hi3dError(f"{what} failed (rc={rc}): {msg}")
