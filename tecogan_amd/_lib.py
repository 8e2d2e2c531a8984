"""ctypes binding of libtecogan_hip.so (the C ABI declared in include/tecogan_hip.h).

The library is the product: if it is missing, or a call fails, this module raises --
there is no CPU or eager fallback anywhere in `tecogan_amd`.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# TECOGAN_HIP_LIB: another build of the same library (A/B of compile-time layout constants, tools/build_variant.py)
LIB_PATH = os.environ.get("TECOGAN_HIP_LIB") or os.path.join(_HERE, "libtecogan_hip.so")

TG_F32, TG_BF16 = 0, 1
ACT_NONE, ACT_RELU, ACT_LRELU, ACT_TANH, ACT_SIGMOID = 0, 1, 2, 3, 4
CONV_COEXIST = 1


class ConvDesc(C.Structure):
    _fields_ = [("N", C.c_int32), ("Hin", C.c_int32), ("Win", C.c_int32), ("Cin", C.c_int32),
                ("Hout", C.c_int32), ("Wout", C.c_int32), ("Cout", C.c_int32),
                ("KH", C.c_int32), ("KW", C.c_int32), ("stride", C.c_int32),
                ("pad_t", C.c_int32), ("pad_l", C.c_int32), ("mode", C.c_int32),
                ("in_dtype", C.c_int32), ("out_dtype", C.c_int32),
                ("act", C.c_int32), ("act_alpha", C.c_float),
                ("mask_act", C.c_int32), ("mask_alpha", C.c_float), ("flags", C.c_int32)]


_P, _I, _L, _F = C.c_void_p, C.c_int, C.c_int64, C.c_float
_D = C.POINTER(ConvDesc)

# name -> argtypes; every function returns int (0 = ok)
SIGNATURES = {
    "tg_conv_forward": [_D, _P, _P, _P, _P, _P, _P, _P],
    "tg_conv_wgrad": [_D, _P, _I, _I, _P, _I, _I, _P, _P, _P],
    "tg_conv_wgrad_grouped": [_D, _I, _P, _I, _I, _P, _I, _I, _P, _P, _P],
    "tg_conv_wgrad_grouped_plus": [_D, _I, _P, _I, _I, _P, _I, _I, _P, _P, _P, _I, _I, _P, _P, _P, _P],
    "tg_conv_wgrad_multi": [_D, _I, _P, _I, _P, _P, _I, _P, _P, _P, _P],
    "tg_colsum": [_P, _I, _L, _I, _P, _P],
    "tg_pack_weights": [_P, _P, _I, _P, _I, _I, _P],
    "tg_pack_weights_both": [_P, _P, _P, _I, _P, _I, _P],
    "tg_warp_s2d_forward": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _F, _P, _P],
    "tg_warp_s2d_backward": [_P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _F, _P],
    "tg_warp_forward": [_P, _P, _P, _I, _I, _I, _I, _P],
    "tg_warp_backward": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "tg_upscale4_forward": [_P, _P, _I, _I, _I, _I, _F, _P],
    "tg_upscale4_backward": [_P, _P, _I, _I, _I, _I, _F, _P],
    "tg_maxpool2_forward": [_P, _P, _I, _I, _I, _I, _I, _P],
    "tg_maxpool2_backward": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _P, _P],
    "tg_upsample2_forward": [_P, _P, _I, _I, _I, _I, _I, _P],
    "tg_upsample2_backward": [_P, _P, _I, _I, _I, _I, _I, _P, _I, _F, _P],
    "tg_bicubic_add_preprocess": [_P, _P, _I, _I, _P, _P, _I, _I, _I, _P],
    "tg_resblock": [_I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "tg_resblock_chain_scratch_bytes": [_I, _I, _I, C.POINTER(C.c_int64)],
    "tg_resblock_chain": [_I, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "tg_resblock_plane_scratch_bytes": [_I, _I, _I, C.POINTER(C.c_int64)],
    "tg_resblock_plane": [_P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "tg_pack_weights_frag": [_P, _P, _P, _P, _I, _P],
    "tg_hr_tail_backward": [_P, _F, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    "tg_deconv_lat_forward": [_P, _P, _P, _P, _I, _I, _I, _P],
    "tg_deconv_lat_backward": [_P, _P, _P, _P, _I, _I, _I, _P],
    "tg_conv3x3_c64_frag": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _P],
    "tg_pack_wide_frag": [_P, _P, _I, _I, _I, _P],
    "tg_conv3x3_wide_frag": [_D, _P, _P, _P, _P, _P, _P, _I, _I, _P],
    "tg_pack_taps_frag": [_P, _P, _I, _I, _I, _P],
    "tg_pack_taps_frag_multi": [_P, _P, _P, _P, _I, _P],
    "tg_conv4x4s2_frag": [_D, _P, _P, _P, _P, _P, _P, _P, _P],
    "tg_resblock_c64_thr": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    "tg_hr_tail_train": [_P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _I, _I, _I, _P],
    "tg_act_backward": [_P, _P, _P, _I, _I, _L, _I, _F, _F, _P],
    "tg_concat2_pad": [_P, _I, _P, _I, _P, _I, _I, _L, _F, _P],
    "tg_lincomb": [_P, _P, _P, _L, _F, _F, _I, _P],
    "tg_affine": [_P, _P, _L, _F, _F, _P],
    "tg_schedule_step": [_P, _P, _I, _I, _P, _F, _F, _F, _P],
    "tg_bn_lrelu_forward": [_P, _P, _I, _L, _I, _P, _F, _F, _P, _P, _I, _P],
    "tg_bn_lrelu_backward": [_P, _P, _P, _P, _I, _L, _I, _P, _F, _F, _P, _P, _I, _P],
    "tg_seq_gather": [_P, _P, _I, _I, _I, _L, _P, _P],
    "tg_adam_tf": [_P, _P, _P, _P, _L, _P, _F, _P],
    "tg_sum_sq_diff": [_P, _P, _I, _L, _F, _P, _P],
    "tg_sum_abs_diff": [_P, _P, _I, _L, _F, _P, _P],
    "tg_pingpong": [_P, _P, _I, _I, _L, _F, _F, _P, _P],
    "tg_vgg_preprocess_forward": [_P, _P, _I, _L, _I, _P],
    "tg_vgg_preprocess_backward": [_P, _I, _P, _L, _I, _P],
    "tg_cosine_loss": [_P, _P, _I, _L, _I, _F, _F, _P, _P, _P],
    "tg_l1_loss": [_P, _P, _I, _L, _F, _F, _P, _P, _P, _P],
    "tg_gan_losses": [_P, _P, _I, _F, _F, _P, _P, _P, _P, _P, _P],
    "tg_dt_ratio": [_P, _F, _F, _F, _P, _P],
    "tg_pack_d_input_forward": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "tg_pack_d_input_backward": [_P, _I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
}



class ProfEntry(C.Structure):
    _fields_ = [("name", C.c_char * 96), ("calls", C.c_int64), ("total_us", C.c_double), ("flops", C.c_double),
                ("bytes", C.c_double)]


SIGNATURES["tg_gauss_down4_preprocess"] = [_P, _P, _P, _I, _I, _I, _I, C.POINTER(C.c_float), _I, _P]
SIGNATURES["tg_frame_to_u8"] = [_P, _P, _L, _I, _P]
SIGNATURES["tg_prof_enable"] = [_I]
SIGNATURES["tg_prof_collect"] = [C.POINTER(ProfEntry), _I, C.POINTER(C.c_int)]
SIGNATURES["tg_prof_stamp"] = [_P, _P]
SIGNATURES["tg_graph_node_count"] = [_P, C.POINTER(C.c_int), C.POINTER(C.c_int)]

_lib = None


class TecoHipError(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle; raises if the HIP library is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise TecoHipError(
                "libtecogan_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `python tecogan_amd/build.py`; tecogan_amd has no fallback path." % LIB_PATH)
        # One HIP runtime per process: PyTorch (device memory, streams, hipGraph capture) bundles its own libamdhip64 and
        # must be loaded BEFORE this library resolves the runtime.  Loaded the other way round (observed with
        # `build()` then `smoke()` in one interpreter) the process ends up with two runtimes and every launch from
        # this library fails with "no ROCm-capable device is detected".
        import torch  # noqa: F401
        h = C.CDLL(LIB_PATH)
        for name, args in SIGNATURES.items():
            fn = getattr(h, name)
            fn.argtypes = args
            fn.restype = C.c_int
        h.tg_last_error_string.restype = C.c_char_p
        h.tg_version.restype = C.c_int
        _lib = h
    return _lib


def check(rc, what):
    if rc != 0:
        raise TecoHipError("%s failed (%d): %s" % (what, rc, lib().tg_last_error_string().decode()))
