"""ctypes binding of libcilrs_hip.so (C-ABI declared in include/cilrs_hip.h).

The library is the product: there is NO CPU or eager-PyTorch fallback.  If the shared object is
missing or a call fails, a RuntimeError is raised with the library's own message.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# CILRS_LIB: another build of the same library (A/B timing of kernel variants on one box)
LIB_PATH = os.environ.get("CILRS_LIB") or os.path.join(_HERE, "libcilrs_hip.so")

c_float_p = C.POINTER(C.c_float)
vp = C.c_void_p
sz = C.c_size_t
i32 = C.c_int
i64 = C.c_int64
f32 = C.c_float
f64 = C.c_double
u64 = C.c_uint64


class Buffers(C.Structure):
    """struct cilrs_buffers"""
    _fields_ = [("params", vp), ("grads", vp), ("bn_running", vp), ("bn_nbt", vp),
                ("workspace", vp)]


class AdamArgs(C.Structure):
    """struct cilrs_adam_args"""
    _fields_ = [("exp_avg", vp), ("exp_avg_sq", vp), ("lr", C.c_double), ("beta1", C.c_double),
                ("beta2", C.c_double), ("eps", C.c_double), ("weight_decay", C.c_double),
                ("step", C.c_int64), ("grad_scale", C.c_float)]


# symbol -> (restype, argtypes); every symbol include/cilrs_hip.h declares is listed here
SIGNATURES = {
    "cilrs_version": (i32, []),
    "cilrs_last_error": (C.c_char_p, []),
    "cilrs_num_params": (i32, []),
    "cilrs_num_bn": (i32, []),
    "cilrs_param_arena_floats": (sz, []),
    "cilrs_param_count": (sz, []),
    "cilrs_param_info": (i32, [i32, C.c_char_p, i32, C.POINTER(sz), C.POINTER(sz),
                               C.POINTER(i32), C.POINTER(i32)]),
    "cilrs_bn_info": (i32, [i32, C.c_char_p, i32, C.POINTER(i32), C.POINTER(sz), C.POINTER(sz)]),
    "cilrs_bn_arena_floats": (sz, []),
    "cilrs_num_variants": (i32, []),
    "cilrs_variant_num_params": (i32, [i32]),
    "cilrs_variant_num_bn": (i32, [i32]),
    "cilrs_variant_param_arena_floats": (sz, [i32]),
    "cilrs_variant_param_count": (sz, [i32]),
    "cilrs_variant_bn_arena_floats": (sz, [i32]),
    "cilrs_variant_feature_width": (i32, [i32]),
    "cilrs_variant_param_info": (i32, [i32, i32, C.c_char_p, i32, C.POINTER(sz), C.POINTER(sz),
                                       C.POINTER(i32), C.POINTER(i32)]),
    "cilrs_variant_bn_info": (i32, [i32, i32, C.c_char_p, i32, C.POINTER(i32), C.POINTER(sz),
                                    C.POINTER(sz)]),
    "cilrs_net_create": (i32, [i32, i32, i32, C.POINTER(vp)]),
    "cilrs_net_create_variant": (i32, [i32, i32, i32, i32, C.POINTER(vp)]),
    "cilrs_net_create_ex": (i32, [i32, i32, i32, i32, C.c_uint, C.POINTER(vp)]),
    "cilrs_net_destroy": (None, [vp]),
    "cilrs_net_workspace_bytes": (sz, [vp]),
    "cilrs_net_status_offset": (sz, [vp]),
    "cilrs_net_set_weights_key": (i32, [vp, u64]),
    "cilrs_net_activation_info": (i32, [vp, i32, C.POINTER(sz), C.POINTER(sz), C.POINTER(sz),
                                        C.POINTER(i32)]),
    "cilrs_dropout": (i32, [vp, i32, i32, i32, f32, u64, i32, vp]),
    "cilrs_net_forward": (i32, [vp, C.POINTER(Buffers), vp, C.c_long, C.c_long, C.c_long,
                                C.c_long, vp, vp, i32, f32, u64, vp, vp, vp]),
    "cilrs_net_forward_u8": (i32, [vp, C.POINTER(Buffers), vp, vp, vp, vp, vp, vp]),
    "cilrs_net_forward_camera": (i32, [vp, C.POINTER(Buffers), vp, i32, i32, i32, C.c_long,
                                       C.c_long, vp, vp, vp, vp, vp]),
    "cilrs_net_forward_u8_f16": (i32, [vp, C.POINTER(Buffers), vp, vp, vp, vp, vp, vp]),
    "cilrs_net_forward_u8_f16_graph": (i32, [vp, C.POINTER(Buffers), vp, vp, vp, vp, vp, vp]),
    "cilrs_net_forward_u8_graph": (i32, [vp, C.POINTER(Buffers), vp, vp, vp, vp, vp, vp]),
    "cilrs_net_forward_u8_b1": (i32, [vp, C.POINTER(Buffers), vp, vp, vp, vp, vp, vp]),
    "cilrs_net_forward_camera_b1": (i32, [vp, C.POINTER(Buffers), vp, i32, i32, i32, C.c_long, vp, vp,
                                          vp, vp, i32, vp]),
    "cilrs_net_forward_u8_b1_post": (i32, [vp, C.POINTER(Buffers), vp, vp, vp, vp, vp, vp, i32, vp]),
    "cilrs_net_forward_u8_b1_sync": (i32, [vp, C.POINTER(Buffers), vp, vp, vp, vp, vp, vp]),
    "cilrs_net_b1_stages": (i32, [vp]),
    "cilrs_net_b1_set_epoch": (i32, [vp, C.POINTER(Buffers), i32, vp]),
    "cilrs_net_wino_convs": (i32, [vp]),
    "cilrs_net_b1_stage_us": (i32, [vp, C.POINTER(Buffers), c_float_p, c_float_p, i32]),
    "cilrs_net_forward_u8_bf16": (i32, [vp, C.POINTER(Buffers), vp, vp, vp, vp, vp, vp]),
    "cilrs_net_forward_u8_bf16_graph": (i32, [vp, C.POINTER(Buffers), vp, vp, vp, vp, vp, vp]),
    "cilrs_loss_fwd_bwd": (i32, [vp, vp, vp, vp, i32, i32, c_float_p, f32, vp, vp, vp, vp]),
    "cilrs_net_backward": (i32, [vp, C.POINTER(Buffers), vp, vp, i32, i32, vp]),
    "cilrs_stem_conv_fwd": (i32, [vp, vp, vp, vp, i32, i32, i32, vp, vp]),
    "cilrs_stem_conv_wgrad_scratch_floats": (sz, [i32, i32, i32]),
    "cilrs_stem_conv_wgrad": (i32, [vp, vp, vp, vp, sz, i32, i32, i32, vp]),
    "cilrs_net_backward_step": (i32, [vp, C.POINTER(Buffers), vp, vp, vp, vp]),
    "cilrs_segment_range": (i32, [i32, C.POINTER(sz), C.POINTER(sz)]),
    "cilrs_variant_segment_range": (i32, [i32, i32, C.POINTER(sz), C.POINTER(sz)]),
    "cilrs_sqnorm_scratch_bytes": (sz, []),
    "cilrs_grad_sqnorm": (i32, [vp, sz, f32, vp, vp, vp]),
    "cilrs_adam_step": (i32, [vp, vp, vp, vp, sz, f64, f64, f64, f64, f64, i64, vp, f32, vp]),
    "cilrs_scale": (i32, [vp, sz, vp, f32, vp]),
    "cilrs_augment_u8": (i32, [vp, vp, i32, i32, i32, vp, vp, vp]),
    "cilrs_eval_acc_doubles": (i32, []),
    "cilrs_eval_accumulate": (i32, [vp, vp, vp, vp, vp, i32, vp, vp, vp]),
    "cilrs_net_profile_enable": (i32, [vp, i32]),
    "cilrs_net_profile_collect": (i32, [vp]),
    "cilrs_net_profile_count": (i32, [vp]),
    "cilrs_net_profile_entry": (i32, [vp, i32, C.c_char_p, i32, C.POINTER(C.c_longlong),
                                      C.POINTER(f64), C.POINTER(f64), C.POINTER(f64)]),
    "cilrs_net_profile_reset": (i32, [vp]),
    "cilrs_conv2d_fwd": (i32, [vp, vp, vp] + [i32] * 11 + [vp, sz, vp]),
    "cilrs_conv2d_dgrad": (i32, [vp, vp, vp, vp] + [i32] * 11 + [vp, sz, vp]),
    "cilrs_conv2d_wgrad_scratch_floats": (sz, [i32] * 9),
    "cilrs_conv2d_wgrad": (i32, [vp, vp, vp, vp] + [i32] * 10 + [vp]),
    "cilrs_conv2d_16_scratch_halfs": (sz, [i32] * 8),
    "cilrs_conv2d_fwd_16": (i32, [vp, vp, vp, vp] + [i32] * 9 + [vp, vp]),
    "cilrs_conv2d_dgrad_16": (i32, [vp, vp, vp, vp] + [i32] * 9 + [vp, vp]),
    "cilrs_conv2d_wgrad_16_scratch_floats": (sz, [i32] * 8),
    "cilrs_conv2d_wgrad_16": (i32, [vp, vp, vp, vp] + [i32] * 9 + [vp, vp]),
    "cilrs_conv2d_train_16": (i32, [vp] * 9 + [i32, vp] + [i32] * 12 + [vp, vp]),
    "cilrs_bn16_train_fwd": (i32, [vp, i32, i32, vp, vp, vp, vp, vp, C.c_float, C.c_float, vp, i32,
                                   vp, vp, vp, i32, vp]),
    "cilrs_bn16_bwd": (i32, [vp, vp, vp, i32, i32, vp, vp, i32, vp, vp, vp, vp, vp, vp, i32, vp]),
    "cilrs_conv2d_wino_split": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, sz, vp, vp, vp]),
    "cilrs_conv2d_wino_scratch_floats": (sz, [i32, i32]),
    "cilrs_conv2d_wino_fwd": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, vp, vp]),
    "cilrs_wino_filter_transform": (i32, [vp, vp, i32, i32, i32, vp]),
    "cilrs_conv2d_wino_stamps": (i32, [vp]),
    "cilrs_conv2d_wino_pre": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "cilrs_conv2d_wino_dgrad": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp]),
    "cilrs_conv2d_wino_wgrad_scratch_floats": (sz, [i32, i32, i32, i32, i32]),
    "cilrs_conv2d_wino_wgrad": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, vp, sz, vp]),
    "cilrs_bn_partial_floats": (sz, [i32]),
    "cilrs_bn_train_fwd": (i32, [vp, i32, i32, vp, vp, vp, vp, vp, f32, f32, vp, i32, vp, vp, vp,
                                 vp]),
    "cilrs_bn_eval_fwd": (i32, [vp, i32, i32, vp, vp, vp, vp, f32, vp, i32, vp, vp, vp]),
    "cilrs_bn_bwd": (i32, [vp, vp, vp, i32, i32, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp]),
    "cilrs_linear_fwd": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "cilrs_linear_bwd": (i32, [vp, vp, vp, vp, f32, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32,
                               vp]),
    "cilrs_maxpool_fwd": (i32, [vp, vp, vp, i32, i32, i32, i32, vp]),
    "cilrs_maxpool_bwd": (i32, [vp, vp, vp, i32, i32, i32, i32, vp]),
}

_lib = None


def lib():
    """Load (once) and return the bound library; raises if the HIP extension is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"CILRS HIP extension not built: {LIB_PATH} is missing. Build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or `make` in csrc/). "
            "There is no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(L, name)          # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def check(rc: int):
    if rc != 0:
        msg = lib().cilrs_last_error()
        raise RuntimeError("cilrs_hip: " + (msg.decode() if msg else f"error {rc}"))


def ptr(t):
    """Device (or host) pointer of a torch tensor, or None."""
    return None if t is None else C.c_void_p(t.data_ptr())
