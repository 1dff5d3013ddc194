"""ctypes binding of include/gddim_amd.h.

The product path has no CPU fallback: if the native library is missing or a call fails, a
`GddimError` is raised.  Nothing here imports the oracle.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# GDDIM_LIB selects another build of the same library (A/B measurements of kernel variants inside one gpurun call)
LIB_PATH = os.environ.get("GDDIM_LIB") or os.path.join(_HERE, "libgddim_amd.so")

GD_MAX_LEVELS = 8


class GddimError(RuntimeError):
    pass


class ModelDesc(ctypes.Structure):
    _fields_ = [
        ("image_size", ctypes.c_int),
        ("in_channels", ctypes.c_int),
        ("nf", ctypes.c_int),
        ("num_res_blocks", ctypes.c_int),
        ("num_levels", ctypes.c_int),
        ("ch_mult", ctypes.c_int * GD_MAX_LEVELS),
        ("num_attn_resolutions", ctypes.c_int),
        ("attn_resolutions", ctypes.c_int * GD_MAX_LEVELS),
        ("conditional", ctypes.c_int),
        ("skip_rescale", ctypes.c_int),
        ("progressive_input_residual", ctypes.c_int),
        ("scale_by_sigma", ctypes.c_int),
        ("centered", ctypes.c_int),
        ("max_batch", ctypes.c_int),
        ("positional_embedding", ctypes.c_int),
        ("no_fir", ctypes.c_int),
    ]


class ProfEntry(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char * 128), ("ms", ctypes.c_double), ("flops", ctypes.c_double),
                ("bytes", ctypes.c_double), ("launches", ctypes.c_longlong)]


_P = ctypes.c_void_p
_I = ctypes.c_int
_LL = ctypes.c_longlong
_F = ctypes.c_float
_D = ctypes.c_double
_SZ = ctypes.c_size_t

# name -> (restype, argtypes); every symbol include/gddim_amd.h declares
SIGNATURES = {
    "gd_last_error": (ctypes.c_char_p, []),
    "gd_version": (_I, []),
    "gd_device_count": (_I, []),
    "gd_options_reload": (_I, []),
    "gd_cld_r_table": (_I, [_D, _D, _D, _D, _D, _I, _D, _SZ, _P]),
    "gd_cld_r_table_f32": (_I, [_D, _D, _D, _D, _D, _I, _D, _SZ, _P]),
    "gd_cld_psi2_table": (_I, [_D, _D, _D, _D, _SZ, _P]),
    "gd_cld_lambda_hat_psi_table": (_I, [_D, _D, _D, _D, _P, _P, _SZ, _D, _SZ, _P]),
    "gd_cld_lambda_cond_rev_cov": (_I, [_D, _D, _D, _D, _P, _P, _SZ, _D, _D, _I, _P]),
    "gd_engine_create": (_I, [ctypes.POINTER(ModelDesc), _I, ctypes.POINTER(_P)]),
    "gd_engine_destroy": (_I, [_P]),
    "gd_engine_plan_json": (_I, [_P, _P, _SZ, ctypes.POINTER(_SZ)]),
    "gd_engine_flops_per_image": (_D, [_P, _I]),
    "gd_engine_num_params": (_I, [_P]),
    "gd_engine_param_info": (_I, [_P, _I, _P, _SZ, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(_I)]),
    "gd_engine_load_param": (_I, [_P, ctypes.c_char_p, _P, ctypes.POINTER(ctypes.c_int64), _I]),
    "gd_engine_params_ready": (_I, [_P]),
    "gd_engine_packed_weight": (_I, [_P, ctypes.c_char_p, _P, _SZ, ctypes.POINTER(_SZ)]),
    "gd_engine_weight_bytes": (_I, [_P, ctypes.POINTER(ctypes.c_longlong)]),
    "gd_engine_set_precision": (_I, [_P, _I]),
    "gd_unet_forward": (_I, [_P, _P, _P, _P, _I, _P]),
    "gd_cld_eps": (_I, [_P, _P, _P, _P, _I, _P]),
    "gd_engine_set_schedule_cld": (_I, [_P, _I, _I, _P, _P, _I, _F, _P, _P, _P]),
    "gd_engine_set_transforms_cld": (_I, [_P, _P, _P, _P, _P, _P]),
    "gd_engine_set_noise_cld": (_I, [_P, _P, _P, _P]),
    "gd_engine_set_seed": (_I, [_P, ctypes.c_ulonglong]),
    "gd_randn_pairs": (_I, [_P, _LL, ctypes.c_ulonglong, ctypes.c_uint, ctypes.c_ulonglong, _P]),
    "gd_sample_cld": (_I, [_P, _I, _P, _P, _P, _I, _I, _P]),
    "gd_engine_set_schedule_blur": (_I, [_P, _I, _P, _P, _P]),
    "gd_engine_set_schedule_blur_deis": (_I, [_P, _I, _I, _P, _P]),
    "gd_engine_set_model2eps_blur": (_I, [_P, _P]),
    "gd_sample_blur": (_I, [_P, _I, _P, _P, _I, _I, _P]),
    "gd_engine_profile": (_I, [_P, _I]),
    "gd_engine_profile_read": (_I, [_P, ctypes.POINTER(ProfEntry), _I, ctypes.POINTER(_I)]),
    "gd_ab_step": (_I, [_P, _P, _P, _P, _P, _P, _I, _LL, _P]),
    "gd_ab_step_scalar": (_I, [_P, _P, _P, _P, _P, _P, _I, _LL, _P]),
    "gd_bmm2x2": (_I, [_P, _P, _P, _LL, _I, _LL, _P]),
    "gd_dct2d_32": (_I, [_P, _P, _I, _I, _I, _P]),
    "gd_conv2d": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F, _F, _P]),
    "gd_conv3x3_winograd2d": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _F, _P]),
    "gd_conv3x3_winograd42": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _F, _P]),
    "gd_conv3x3_winograd42_weights": (_I, [_P, _P, _I, _I, _P]),
    "gd_conv3x3_winograd42_pretransformed": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _F, _P]),
    "gd_conv3x3_winograd2d_split6_weight_bytes": (_LL, [_I, _I]),
    "gd_conv3x3_winograd2d_split6_weights": (_I, [_P, _P, _I, _I, _P]),
    "gd_conv3x3_winograd2d_split6": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _P]),
    "gd_conv3x3_winograd2d_split6_gn": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _P]),
    "gd_conv3x3_winograd42_gn": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _F, _P]),
    "gd_group_norm": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "gd_fir_resample": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "gd_attention": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "gd_attention_fused": (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    "gd_attention_block_fused": (_I, [_P, _P, _P, _P, _P, _F, _P, _P, _I, _I, _I, _P]),
}

_lib = None


def lib():
    """Load libgddim_amd.so (built by gddim_amd/build.py).  Raises GddimError when absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GddimError("native library %s not found: run `python -m gddim_amd.build` "
                             "(no CPU fallback exists)" % LIB_PATH)
        # torch first: its wheel bundles a HIP runtime of its own, and whichever libamdhip64 is loaded first serves the whole process.
        # Loading this library (linked against /opt/rocm) before torch left torch without a device (build() followed by smoke() in
        # one process).
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        try:
            l = ctypes.CDLL(LIB_PATH)
        except OSError as ex:
            raise GddimError("cannot load %s: %s" % (LIB_PATH, ex))
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(l, name)
            except AttributeError:
                raise GddimError("symbol %s missing from %s" % (name, LIB_PATH))
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc):
    if rc != 0:
        raise GddimError(lib().gd_last_error().decode(errors="replace"))


def reload_options():
    """Re-read the GDDIM_* / GD_* environment into the library's option table (csrc/options.h); engines re-read it when they are created."""
    check(lib().gd_options_reload())


def device_count():
    return lib().gd_device_count()
