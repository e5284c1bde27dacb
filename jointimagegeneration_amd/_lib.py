"""ctypes binding of libguidegen_hip.so (the C-ABI in include/guidegen_hip.h).

There is NO fallback: if the shared library is missing or a symbol is absent this module raises, and every
product op raises RuntimeError when a call returns a negative gg_status (message from gg_last_error()).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libguidegen_hip.so")

GG_BF16, GG_F32 = 0, 1

vp, i32, i64, u64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_float


class ConvDesc(C.Structure):
    _fields_ = [
        ("N", i32), ("D", i32), ("H", i32), ("W", i32),
        ("C1", i32), ("C2", i32), ("Cout", i32), ("Cout_pad", i32),
        ("kd", i32), ("kh", i32), ("kw", i32), ("stride", i32), ("pad", i32), ("upsample", i32),
        ("Do", i32), ("Ho", i32), ("Wo", i32), ("out_dtype", i32), ("prologue_act", i32), ("path_hint", i32),
        ("src1", vp), ("src2", vp), ("weight", vp), ("bias", vp), ("bias_stride", i64),
        ("residual", vp), ("out", vp), ("gn_scale", vp), ("gn_shift", vp), ("workspace", vp), ("workspace_bytes", i64), ("reserved_ptr", vp), ("gn_acc", vp),
        ("ddim_x", vp), ("ddim_scalars", vp), ("ddim_pred_x0", vp), ("ddim_unet_in", vp), ("ddim_unet_in_stride", i64),
        ("epilogue_geglu", i32), ("pro_c_logical", i32), ("pro_acc1", vp), ("pro_acc2", vp), ("pro_gamma", vp), ("pro_beta", vp),
        ("pro_eps", f32), ("skip_C1", i32), ("skip_C2", i32), ("reserved_tail", i32), ("skip_src1", vp), ("skip_src2", vp), ("skip_weight", vp),
        ("post_xt", vp), ("post_labels_out", vp), ("post_scalars", vp), ("post_E", vp), ("post_philox_seed", C.c_uint64),
        ("post_philox_offset_dev", vp), ("post_onehot_out", vp), ("post_onehot_stride", i64), ("post_draw", i32), ("reserved_tail2", i32),
    ]


class AttentionDesc(C.Structure):
    _fields_ = [
        ("N", i32), ("heads", i32), ("head_dim", i32), ("Tq", i32), ("Tkv", i32),
        ("ldq", i64), ("hsq", i64), ("ldk", i64), ("hsk", i64), ("ldv", i64), ("hsv", i64), ("ldo", i64), ("hso", i64),
        ("scale", f32), ("reserved", i32),
        ("q", vp), ("k", vp), ("v", vp), ("out", vp), ("workspace", vp), ("workspace_bytes", i64),
    ]


# name -> (restype, argtypes); must list every symbol declared in include/guidegen_hip.h
SIGNATURES = {
    "gg_last_error": (C.c_char_p, []),
    "gg_version": (C.c_int, []),
    "gg_conv_packed_weight_bytes": (i64, [i32, i32, i32]),
    "gg_conv_pack_weight": (C.c_int, [vp, i32, i32, i32, i32, vp, vp]),
    "gg_conv_forward": (C.c_int, [C.POINTER(ConvDesc), vp]),
    "gg_conv_workspace_bytes": (i64, [C.POINTER(ConvDesc)]),
    "gg_conv_fuses_prologue": (C.c_int, [C.POINTER(ConvDesc)]),
    "gg_conv_runs_halo_tile": (C.c_int, [C.POINTER(ConvDesc)]),
    "gg_conv_emits_stats": (C.c_int, [C.POINTER(ConvDesc)]),
    "gg_conv_prologue_from_acc": (C.c_int, [C.POINTER(ConvDesc)]),
    "gg_conv_fuses_skip": (C.c_int, [C.POINTER(ConvDesc)]),
    "gg_conv_fuses_ddim": (C.c_int, [C.POINTER(ConvDesc)]),
    "gg_conv_fuses_posterior": (C.c_int, [C.POINTER(ConvDesc)]),
    "gg_groupnorm_workspace_bytes": (i64, [i32, i64, i32]),
    "gg_groupnorm_stats": (C.c_int, [vp, i32, vp, i32, i32, i64, i32, vp, vp, f32, vp, vp, vp, i64, vp]),
    "gg_groupnorm_apply": (C.c_int, [vp, i32, vp, i32, i32, i64, vp, vp, i32, vp, vp]),
    "gg_groupnorm_apply_acc": (C.c_int, [vp, i32, vp, vp, i32, vp, i32, i64, i32, vp, vp, f32, i32, vp, vp]),
    "gg_groupnorm_fused_supported": (C.c_int, [i64, i32, i32, i32]),
    "gg_groupnorm_fused": (C.c_int, [vp, i32, vp, i32, i32, i64, i32, vp, vp, f32, i32, vp, vp]),
    "gg_groupnorm_scale_shift_acc": (C.c_int, [vp, i32, i32, vp, i32, i32, i32, i64, i32, vp, vp, f32, vp, vp, vp]),
    "gg_attention_forward": (C.c_int, [C.POINTER(AttentionDesc), vp]),
    "gg_attention_workspace_bytes": (i64, [C.POINTER(AttentionDesc)]),
    "gg_layernorm": (C.c_int, [vp, i64, i32, vp, vp, f32, vp, vp]),
    "gg_geglu": (C.c_int, [vp, i64, i32, vp, vp]),
    "gg_add": (C.c_int, [vp, vp, i64, vp, vp]),
    "gg_linear_f32": (C.c_int, [vp, i32, i32, vp, vp, i32, i32, vp, i64, vp]),
    "gg_timestep_embedding": (C.c_int, [vp, i32, i32, f32, vp, vp]),
    "gg_nchw_f32_to_cl_bf16": (C.c_int, [vp, i32, i32, i64, vp, i32, i32, i32, vp]),
    "gg_cl_to_nchw_f32": (C.c_int, [vp, i32, i32, i32, i64, i32, vp, vp]),
    "gg_ccdm_posterior_sample": (C.c_int, [vp, i32, i32, vp, vp, u64, vp, i32, vp, i32, i64, vp, vp, vp, i32, vp]),
    "gg_labels_to_onehot": (C.c_int, [vp, i64, i32, vp, i32, vp]),
    "gg_ddim_step": (C.c_int, [vp, vp, i32, vp, vp, i64, i32, vp, vp, i32, vp]),
    "gg_minmax_normalise": (C.c_int, [vp, i64, vp, vp, vp]),
    "gg_ddpm_step": (C.c_int, [vp, vp, i32, vp, vp, i64, i32, vp, i32, vp]),
    "gg_lincomb4": (C.c_int, [vp, vp, vp, vp, f32, f32, f32, f32, f32, i64, vp, vp]),
    "gg_conv_forward_f32": (C.c_int, [C.POINTER(ConvDesc), vp]),
    "gg_groupnorm_f32": (C.c_int, [vp, i32, vp, i32, i32, i64, i32, vp, vp, f32, i32, vp, vp, vp]),
    "gg_attention_forward_f32": (C.c_int, [C.POINTER(AttentionDesc), vp]),
    "gg_mask_to_cond_slice": (C.c_int, [vp, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp, i32, vp, vp]),
    "gg_ubench_mfma_bf16": (C.c_int, [i32, i32, i32, vp, C.POINTER(C.c_double), vp]),
    "gg_ubench_stream_copy": (C.c_int, [vp, vp, i64, vp]),
}

_lib = None


def load() -> C.CDLL:
    """Load the library once; raises (never falls back) if it is absent or incomplete."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the HIP extension is mandatory (there is no CPU fallback). "
            "Build it with `python -c 'import __graft_entry__ as g; g.build()'` or jointimagegeneration_amd/csrc/build.sh")
    # torch FIRST: its wheel bundles its own HIP runtime (torch/lib/libamdhip64.so); loaded before this library, the dynamic loader binds
    # libguidegen_hip.so's `libamdhip64.so.7` to that same copy.  The other order leaves TWO HIP runtimes in the process, and launches
    # through the second one fail with "no ROCm-capable device is detected" on tensors the first one owns.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name, None)
        if fn is None:
            raise RuntimeError(f"libguidegen_hip.so does not export {name}")
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().gg_last_error()
        raise RuntimeError(f"{what} failed with gg_status {rc}: {msg.decode() if msg else ''}")
