"""ctypes binding of libhvc_hip.so (the C ABI declared in include/hvc_hip.h).

The library is the product path: there is no CPU or eager-PyTorch fallback.  Importing this
module succeeds without a GPU (so the ABI can be inspected), but any compute call on a machine
without the built library, or with non-HIP tensors, raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "lib", "libhvc_hip.so")

HVC_F32 = 0
HVC_BF16 = 1
ABI_VERSION = 1

_i, _i64, _u64, _f, _p = C.c_int, C.c_int64, C.c_uint64, C.c_float, C.c_void_p

# name -> (restype, argtypes); mirrors include/hvc_hip.h declaration by declaration.
SIGNATURES = {
    "hvc_abi_version": (_i, []),
    "hvc_last_error": (C.c_char_p, []),
    "hvc_set_seed_counter": (_i, [_p]),
    "hvc_seed_counter_advance": (_i, [_p, C.c_uint32, _p]),
    "hvc_clear_seed_counter_if": (_i, [_p]),
    "hvc_set_option": (_i, [C.c_char_p, _i]),
    "hvc_get_option": (_i, [C.c_char_p, C.POINTER(_i)]),
    "hvc_device_info": (_i, [C.POINTER(_i), C.POINTER(_i), C.c_char_p, _i]),
    "hvc_attention_fwd": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i] + [_i64] * 12 + [_f, _f, _u64, _i, _p]),
    "hvc_attention_fwd_fp8_workspace": (_i64, [_i, _i, _i, _i]),
    "hvc_attention_fwd_fp8": (_i, [_p] * 6 + [_i] * 5 + [_i64] * 12 + [_f, _f, _u64, _p]),
    "hvc_attention_bwd_workspace": (_i64, [_i, _i, _i, _i, _i]),
    "hvc_attention_bwd": (_i, [_p] * 10 + [_i] * 5 + [_i64] * 12 + [_f, _f, _u64, _i, _i, _p]),
    "hvc_gemm": (_i, [_p, _p, _p, _i, _i, _i, _i64, _i64, _i64, _i, _i, _f, _p, _i, _p, _p, _i64, _p, _p, _i64, _i, _i,
                      _f, _u64, _p, _i64, _i, _i, _p]),
    "hvc_gemm_workspace": (_i64, [_i, _i, _i]),
    "hvc_layernorm_fwd": (_i, [_p] * 8 + [_i, _i, _i, _f, _i, _p]),
    "hvc_layernorm_bwd_workspace": (_i64, [_i, _i, _i]),
    "hvc_layernorm_bwd": (_i, [_p] * 14 + [_i, _i, _i, _i, _p]),
    "hvc_branch_bwd_workspace": (_i64, [_i, _i, _i]),
    "hvc_branch_bwd": (_i, [_p] * 7 + [_i, _i, _i, _f, _u64, _i, _p]),
    "hvc_colsum_workspace": (_i64, [_i, _i]),
    "hvc_colsum": (_i, [_p, _p, _p, _i, _i, _i, _p]),
    "hvc_cast": (_i, [_p, _p, _i64, _i, _i, _p]),
    "hvc_im2col": (_i, [_p, _p] + [_i] * 13 + [_i64, _i, _p]),
    "hvc_col2im": (_i, [_p, _p] + [_i] * 13 + [_i64, _i, _p]),
    "hvc_conv_c1_fwd": (_i, [_p, _p, _p, _p] + [_i] * 6 + [_p]),
    "hvc_conv_c1_dw_workspace": (_i64, [_i] * 6),
    "hvc_conv_c1_dw": (_i, [_p, _p, _p, _p] + [_i] * 6 + [_p]),
    "hvc_conv_c1_dx": (_i, [_p, _p, _p] + [_i] * 5 + [_p]),
    "hvc_conv3_halo": (_i, [_p, _p, _p, _p] + [_i] * 6 + [_p]),
    "hvc_conv_o1_fwd": (_i, [_p, _p, _p, _p, _i64, _i, _p]),
    "hvc_conv_o1_bwd_workspace": (_i64, [_i64, _i]),
    "hvc_conv_o1_bwd": (_i, [_p] * 6 + [_i64, _i, _p]),
    "hvc_conv_gemm": (_i, [_i, _p, _p, _p] + [_i] * 14 + [_i64, _i64, _p, _p, _i64, _i, _p, _i64, _i, _i, _p]),
    "hvc_conv_dx_class_columns": (_i, [_i] * 10 + [_p, _p]),
    "hvc_conv_dx_class": (_i, [_p, _p, _p] + [_i] * 19 + [_i64, _i, _p]),
    "hvc_trilinear_fwd": (_i, [_p, _p] + [_i] * 8 + [_p]),
    "hvc_trilinear_bwd_workspace": (_i64, [_i] * 7),
    "hvc_trilinear_bwd": (_i, [_p, _p, _p] + [_i] * 8 + [_p]),
    "hvc_norm_workspace": (_i64, [_i, _i, _i, _i]),
    "hvc_groupnorm_act_fwd": (_i, [_p] * 6 + [_i, _i, _i, _i, _f, _i, _i, _p]),
    "hvc_groupnorm_act_bwd": (_i, [_p] * 9 + [_i, _i, _i, _i, _i, _i, _p]),
    "hvc_bn_relu_pool_fwd": (_i, [_p] * 9 + [_i] * 8 + [_f, _f, _i, _p]),
    "hvc_bn_relu_pool_bwd": (_i, [_p] * 10 + [_i] * 8 + [_i, _p]),
    "hvc_ssim_l1_workspace": (_i64, [_i, _i, _i, _i]),
    "hvc_ssim_l1_fwd": (_i, [_p] * 5 + [_i] * 5 + [_f, _f, _p]),
    "hvc_ssim_l1_bwd": (_i, [_p] * 6 + [_i] * 5 + [_f, _f, _p]),
    "hvc_tv3d_workspace": (_i64, [_i, _i, _i, _i]),
    "hvc_tv3d_fwd": (_i, [_p] * 3 + [_i] * 4 + [_f, _p]),
    "hvc_tv3d_bwd": (_i, [_p] * 3 + [_i] * 4 + [_f, _p]),
    "hvc_spectral_l1_workspace": (_i64, [_i, _i, _i, _i]),
    "hvc_spectral_l1_fwd": (_i, [_p] * 4 + [_i] * 4 + [_p]),
    "hvc_spectral_l1_bwd": (_i, [_p] * 4 + [_i] * 4 + [_p]),
    "hvc_resize_loss_workspace": (_i64, [_i, _i, _i]),
    "hvc_resize_loss_fwd": (_i, [_p] * 4 + [_i] * 5 + [_i64, _i, _i, _p]),
    "hvc_resize_loss_grad": (_i, [_p] * 4 + [_i] * 5 + [_i64, _i, _i, _p]),
    "hvc_view_mean_gap_workspace": (_i64, [_i, _i, _i]),
    "hvc_view_mean_gap_fwd": (_i, [_p] * 4 + [_i] * 5 + [_p]),
    "hvc_view_mean_gap_bwd": (_i, [_p] * 3 + [_i] * 5 + [_p]),
    "hvc_drr_fwd": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _f, _f, _f, _i, _i, _p]),
    "hvc_drr_bwd": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _f, _f, _f, _i, _i, _p]),
}

_lib = None


class HvcLibraryError(RuntimeError):
    pass


def load():
    """Load (once) and return the ctypes handle; raises HvcLibraryError if the .so is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HvcLibraryError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C hybrid-vit-cascade_amd/csrc` (hipcc --offload-arch=gfx950). "
            "There is no CPU fallback for the HVC ops.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = ABI mismatch, fail loudly
        fn.restype = res
        fn.argtypes = args
    if lib.hvc_abi_version() != ABI_VERSION:
        raise HvcLibraryError(f"ABI version mismatch: library {lib.hvc_abi_version()} != binding {ABI_VERSION}")
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().hvc_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what} failed (rc={rc}): {msg}")
