"""The structs of include/wg_rasterizer.h as ctypes types -- pure ctypes, no torch: `_C.py` binds the library with them, and
tests/test_host_cpu.py loads this file on its own (also under AddressSanitizer, where importing torch is not an option)."""
import ctypes as C

_ALLOC_FN = C.CFUNCTYPE(C.c_void_p, C.c_size_t, C.c_void_p)   # wg_alloc_fn
_vp, _i, _f = C.c_void_p, C.c_int, C.c_float


class _ShTone(C.Structure):  # include/wg_rasterizer.h: wg_sh_tone
    _fields_ = [("mul", _vp), ("offset", _vp), ("pre_clamp_max", _f), ("post_clamp_max", _f), ("dL_dmul", _vp), ("dL_doffset", _vp)]


class _SecondImage(C.Structure):  # wg_second_image
    _fields_ = [("colors_precomp2", _vp), ("out_color2", _vp), ("dL_dpix2", _vp), ("dL_dcolor2", _vp)]


class _RawGaussians(C.Structure):  # wg_raw_gaussians
    _fields_ = [("filter_3D", _vp), ("raw_opacities", _vp)]


class _RecolorParent(C.Structure):  # wg_recolor_parent
    _fields_ = [("geom_buffer", _vp), ("binning_buffer", _vp), ("image_buffer", _vp), ("R", _i)]


class _CallOptions(C.Structure):  # wg_call_options
    _fields_ = [("exact_compositing", _i), ("deterministic_backward", _i), ("grad_record", _i)]


class _ForwardArgs(C.Structure):  # wg_forward_args
    _fields_ = ([("struct_size", C.c_size_t), ("geometry_alloc", _ALLOC_FN), ("geometry_user", _vp), ("binning_alloc", _ALLOC_FN), ("binning_user", _vp),
                 ("image_alloc", _ALLOC_FN), ("image_user", _vp)] +
                [(n, _i) for n in ("P", "D", "M", "width", "height", "prefiltered", "debug")] +
                [(n, _f) for n in ("scale_modifier", "tan_fovx", "tan_fovy", "kernel_size")] +
                [(n, _vp) for n in ("background", "means3D", "shs", "colors_precomp", "opacities", "scales", "rotations", "cov3D_precomp", "viewmatrix",
                                    "projmatrix", "cam_pos", "subpixel_offset", "out_color", "radii", "stream")] +
                [("tone", C.POINTER(_ShTone)), ("tone2", C.POINTER(_ShTone)), ("sh_second", _i), ("second", C.POINTER(_SecondImage)),
                 ("raw", C.POINTER(_RawGaussians)), ("recolor", C.POINTER(_RecolorParent)), ("binning_capacity", _i), ("options", C.POINTER(_CallOptions))])


class _BackwardArgs(C.Structure):  # wg_backward_args
    _fields_ = ([("struct_size", C.c_size_t)] + [(n, _i) for n in ("P", "D", "M", "R", "width", "height", "debug")] +
                [(n, _f) for n in ("scale_modifier", "tan_fovx", "tan_fovy", "kernel_size")] +
                [(n, _vp) for n in ("background", "means3D", "shs", "colors_precomp", "scales", "rotations", "cov3D_precomp", "viewmatrix", "projmatrix",
                                    "campos", "subpixel_offset", "radii", "geom_buffer", "binning_buffer", "image_buffer", "dL_dpix", "dL_dmean2D",
                                    "dL_dconic", "dL_dopacity", "dL_dcolor", "dL_dmean3D", "dL_dcov3D", "dL_dsh", "dL_dscale", "dL_drot", "stream")] +
                [("tone", C.POINTER(_ShTone)), ("tone2", C.POINTER(_ShTone)), ("sh_second", _i), ("second", C.POINTER(_SecondImage)),
                 ("raw", C.POINTER(_RawGaussians)), ("options", C.POINTER(_CallOptions))])
