"""ctypes binding of libestd_hip.so (the C ABI declared in include/estd_hip.h).

There is NO fallback: if the library is missing or a call returns a negative status a
RuntimeError is raised (the same exception type ATen shape/device errors surface as in the
reference).  Build with ``python -m estdepth_amd.build`` (or ``__graft_entry__.build()``).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ESTD_LIB", os.path.join(_HERE, "lib", "libestd_hip.so"))

c_float_p = ctypes.c_void_p      # device pointers travel as integers
c_stream = ctypes.c_void_p


class Conv3dDesc(ctypes.Structure):
    """Mirror of struct estd_conv3d_desc (include/estd_hip.h)."""
    _fields_ = [
        ("N", ctypes.c_int), ("D", ctypes.c_int), ("H", ctypes.c_int), ("W", ctypes.c_int),
        ("cin_main", ctypes.c_int), ("in_stride", ctypes.c_int), ("n_tiles", ctypes.c_int),
        ("in_main", ctypes.c_void_p), ("in_extra", ctypes.c_void_p),
        ("w_main", ctypes.c_void_p), ("w_extra", ctypes.c_void_p), ("w_xout", ctypes.c_void_p),
        ("scale", ctypes.c_void_p), ("shift", ctypes.c_void_p),
        ("act_a", ctypes.c_int), ("act_b", ctypes.c_int), ("act_split", ctypes.c_int),
        ("out_main", ctypes.c_void_p), ("out_stride", ctypes.c_int), ("out_channels", ctypes.c_int),
        ("residual", ctypes.c_void_p), ("residual2", ctypes.c_void_p), ("out_scale", ctypes.c_float), ("accumulate", ctypes.c_int),
        ("out_extra", ctypes.c_void_p),
        ("head_w", ctypes.c_void_p), ("head_b", ctypes.c_void_p), ("out_head", ctypes.c_void_p),
        ("stats_partials", ctypes.c_void_p),
        ("w_split", ctypes.c_void_p),
        ("w_wino", ctypes.c_void_p),
        ("w_wino2", ctypes.c_void_p),
        ("gate_r", ctypes.c_void_p), ("gate_stats", ctypes.c_void_p), ("gate_gamma", ctypes.c_void_p), ("gate_beta", ctypes.c_void_p),
    ]


class WarpVolumeOpts(ctypes.Structure):
    """Mirror of struct estd_warp_volume_opts (include/estd_hip.h)."""
    _fields_ = [("depth_per_voxel", ctypes.c_int), ("use_disp", ctypes.c_int), ("border", ctypes.c_int),
                ("depth_min", ctypes.c_float), ("depth_interval", ctypes.c_float),
                ("disp_min", ctypes.c_float), ("disp_interval", ctypes.c_float), ("padding_value", ctypes.c_float)]


class Conv2dDesc(ctypes.Structure):
    """Mirror of struct estd_conv2d_desc (include/estd_hip.h)."""
    _fields_ = [
        ("N", ctypes.c_int), ("H", ctypes.c_int), ("W", ctypes.c_int), ("cin", ctypes.c_int), ("cout", ctypes.c_int),
        ("dilation", ctypes.c_int), ("group_tiles", ctypes.c_int),
        ("in_", ctypes.c_void_p), ("w", ctypes.c_void_p), ("scale", ctypes.c_void_p), ("shift", ctypes.c_void_p),
        ("relu_before_residual", ctypes.c_int), ("relu_after_residual", ctypes.c_int),
        ("residual", ctypes.c_void_p), ("out", ctypes.c_void_p), ("w_split", ctypes.c_void_p), ("w_wino", ctypes.c_void_p),
    ]


class Conv1x1Desc(ctypes.Structure):
    """Mirror of struct estd_conv1x1_desc (include/estd_hip.h)."""
    _fields_ = [
        ("N", ctypes.c_int), ("H", ctypes.c_int), ("W", ctypes.c_int), ("cin", ctypes.c_int), ("cout", ctypes.c_int),
        ("stride", ctypes.c_int), ("relu", ctypes.c_int),
        ("in_", ctypes.c_void_p), ("w", ctypes.c_void_p), ("scale", ctypes.c_void_p), ("shift", ctypes.c_void_p),
        ("residual", ctypes.c_void_p), ("out", ctypes.c_void_p),
    ]


class Conv2dTapsDesc(ctypes.Structure):
    """Mirror of struct estd_conv2d_taps_desc (include/estd_hip.h)."""
    _fields_ = [
        ("N", ctypes.c_int), ("H", ctypes.c_int), ("W", ctypes.c_int), ("cin", ctypes.c_int), ("cout", ctypes.c_int),
        ("ksize", ctypes.c_int), ("stride", ctypes.c_int), ("pad", ctypes.c_int), ("relu", ctypes.c_int),
        ("in_", ctypes.c_void_p), ("w", ctypes.c_void_p), ("scale", ctypes.c_void_p), ("shift", ctypes.c_void_p),
        ("residual", ctypes.c_void_p), ("out", ctypes.c_void_p),
    ]


_SIGNATURES = {
    "estd_version": (ctypes.c_int, []),
    "estd_status_string": (ctypes.c_char_p, [ctypes.c_int]),
    "estd_profile_mark": (ctypes.c_int, [ctypes.c_int, c_stream]),
    "estd_set_reserved_cus": (ctypes.c_int, [ctypes.c_int]),
    "estd_get_reserved_cus": (ctypes.c_int, []),
    "estd_cam_pair_proj": (ctypes.c_int, [c_float_p, c_float_p, c_float_p, c_stream]),
    "estd_cam_sweep_proj": (ctypes.c_int, [c_float_p, c_float_p, c_float_p, c_float_p, c_stream]),
    "estd_cam_volume_mats": (ctypes.c_int, [c_float_p, c_float_p, c_float_p, c_float_p, c_stream]),
    "estd_homo_warping": (ctypes.c_int, [c_float_p, c_float_p, c_float_p, c_float_p,
                                         ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_stream]),
    "estd_homo_warping_px": (ctypes.c_int, [c_float_p, c_float_p, c_float_p, c_float_p,
                                            ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_stream]),
    "estd_warp_volume_ex": (ctypes.c_int, [c_float_p, c_float_p, c_float_p, ctypes.POINTER(WarpVolumeOpts), c_float_p,
                                           ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_stream]),
    "estd_mix1x1_chw_to_hwc": (ctypes.c_int, [c_float_p, c_float_p, c_float_p, c_float_p,
                                              ctypes.c_int, ctypes.c_int, ctypes.c_int, c_stream]),
    "estd_homo_warp_costvol": (ctypes.c_int, [c_float_p, c_float_p, c_float_p, c_float_p, c_float_p,
                                              ctypes.c_int, ctypes.c_int, ctypes.c_int, c_stream]),
    "estd_conv3d_k3": (ctypes.c_int, [ctypes.POINTER(Conv3dDesc), c_stream]),
    "estd_conv3d_k3_split": (ctypes.c_int, [ctypes.POINTER(Conv3dDesc), c_stream]),
    "estd_conv3d_k3_wino": (ctypes.c_int, [ctypes.POINTER(Conv3dDesc), c_stream]),
    "estd_conv3d_k3_wino2": (ctypes.c_int, [ctypes.POINTER(Conv3dDesc), c_stream]),
    "estd_conv3d_k3_wino2x": (ctypes.c_int, [ctypes.POINTER(Conv3dDesc), c_stream]),
    "estd_conv3d_k3_wino3": (ctypes.c_int, [ctypes.POINTER(Conv3dDesc), c_stream]),
    "estd_conv3d_k3_xout": (ctypes.c_int, [ctypes.POINTER(Conv3dDesc), c_stream]),
    "estd_conv2d_k3": (ctypes.c_int, [ctypes.POINTER(Conv2dDesc), c_stream]),
    "estd_conv2d_k3_split": (ctypes.c_int, [ctypes.POINTER(Conv2dDesc), c_stream]),
    "estd_conv2d_k3_wino": (ctypes.c_int, [ctypes.POINTER(Conv2dDesc), c_stream]),
    "estd_conv2d_k3_wino2": (ctypes.c_int, [ctypes.POINTER(Conv2dDesc), c_stream]),
    "estd_conv3d_k3_grid": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "estd_groupnorm_finalize": (ctypes.c_int, [c_float_p, ctypes.c_int, ctypes.c_double, ctypes.c_float,
                                               c_float_p, c_stream]),
    "estd_softargmin_up": (ctypes.c_int, [c_float_p, c_float_p, c_float_p, c_float_p,
                                          ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_stream]),
    "estd_warp_volume": (ctypes.c_int, [c_float_p, c_float_p, c_float_p, ctypes.c_float, ctypes.c_float, c_float_p,
                                        ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_stream]),
    "estd_warp_attention": (ctypes.c_int, [c_float_p, ctypes.POINTER(ctypes.c_void_p), c_float_p, ctypes.c_int,
                                           c_float_p, ctypes.c_float, ctypes.c_float, c_float_p,
                                           ctypes.c_int, ctypes.c_int, ctypes.c_int, c_stream]),
    "estd_attention_prewarped": (ctypes.c_int, [c_float_p, ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, c_float_p,
                                                ctypes.c_int64, c_stream]),
    "estd_gru_reset_apply": (ctypes.c_int, [c_float_p, c_float_p, c_float_p, c_float_p, c_float_p, c_float_p,
                                            ctypes.c_int64, c_stream]),
    "estd_gru_blend": (ctypes.c_int, [c_float_p] * 10 + [ctypes.c_int, ctypes.c_int64, c_stream]),
    "estd_bn_act_nhwc": (ctypes.c_int, [c_float_p, c_float_p, c_float_p, c_float_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int, c_stream]),
    "estd_conv1x1_nhwc": (ctypes.c_int, [ctypes.POINTER(Conv1x1Desc), c_stream]),
    "estd_conv2d_taps_nhwc": (ctypes.c_int, [ctypes.POINTER(Conv2dTapsDesc), c_stream]),
    "estd_stem7x7s2_nhwc": (ctypes.c_int, [c_float_p, c_float_p, c_float_p, c_float_p, c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_stream]),
    "estd_maxpool3x3s2_nhwc": (ctypes.c_int, [c_float_p, c_float_p] + [ctypes.c_int] * 4 + [c_stream]),
    "estd_avgpool_nhwc": (ctypes.c_int, [c_float_p, c_float_p] + [ctypes.c_int] * 5 + [c_stream]),
    "estd_conv2d_small_nhwc": (ctypes.c_int, [c_float_p, c_float_p, c_float_p, c_float_p, c_float_p] + [ctypes.c_int] * 8 + [c_stream]),
    "estd_conv2d_k3_to16_nhwc": (ctypes.c_int, [c_float_p, c_float_p, c_float_p, c_float_p, c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                ctypes.c_int, ctypes.c_int, c_stream]),
    "estd_normalise_nhwc": (ctypes.c_int, [c_float_p, c_float_p, ctypes.c_int, ctypes.c_int64, c_stream]),
    "estd_stem3x3s2_nhwc": (ctypes.c_int, [c_float_p, c_float_p, c_float_p, c_float_p, c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_stream]),
    "estd_nhwc_to_planes": (ctypes.c_int, [c_float_p, ctypes.c_int, c_float_p, ctypes.c_int, ctypes.c_int64, c_stream]),
    "estd_planes_cat_nhwc": (ctypes.c_int, [c_float_p, ctypes.c_int, c_float_p, ctypes.c_int, ctypes.c_int, c_float_p, ctypes.c_int, ctypes.c_int64, c_stream]),
    "estd_upsample2_cat_nhwc": (ctypes.c_int, [c_float_p, ctypes.c_int, c_float_p, ctypes.c_int, c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_stream]),
    "estd_disp_head_nhwc": (ctypes.c_int, [c_float_p, c_float_p, c_float_p, ctypes.c_float, c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                           ctypes.c_int, c_stream]),
    "estd_spp_upsample_cat": (ctypes.c_int, [c_float_p, ctypes.c_int, c_float_p, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p),
                                             ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.c_int,
                                             c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_stream]),
    "estd_cdhw_to_vol": (ctypes.c_int, [c_float_p, c_float_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int, ctypes.c_int, c_stream]),
    "estd_vol_to_cdhw": (ctypes.c_int, [c_float_p, c_float_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int, ctypes.c_int, c_stream]),
}

# the superseded A/B kernels: exported only by a library built with ESTD_BUILD_AB=1 (estdepth_amd/build.py)
AB_SYMBOLS = ("estd_conv3d_k3_split", "estd_conv3d_k3_wino", "estd_conv2d_k3_split", "estd_conv2d_k3_wino", "estd_conv3d_k3_wino2x")
EXPORTED_SYMBOLS = tuple(k for k in _SIGNATURES if k not in AB_SYMBOLS)

_lib = None


def lib():
    """Load the shared library once.  Raises RuntimeError when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libestd_hip.so not found at %s -- build it with `python -m estdepth_amd.build` "
                               "(there is no CPU/eager fallback)" % LIB_PATH)
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            if name in AB_SYMBOLS and not hasattr(handle, name):
                continue
            fn = getattr(handle, name)      # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def has_ab():
    """True when the loaded library carries the superseded A/B kernels (built with ESTD_BUILD_AB=1)."""
    return all(hasattr(lib(), n) for n in AB_SYMBOLS)


def require_ab(what):
    if not has_ab():
        raise RuntimeError("%s needs the A/B kernels: rebuild the library with ESTD_BUILD_AB=1 (python -m estdepth_amd.build)" % what)


def check(status, what):
    if status != 0:
        msg = lib().estd_status_string(status).decode()
        raise RuntimeError("%s failed: %s (estd_status %d)" % (what, msg, status))
