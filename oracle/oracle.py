"""ctypes front-end of the CPU oracle (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product path (wild-gaussians_amd/) never does.  See the header of oracle/wg_oracle.c for what the
oracle restates and how it is pinned (against outputs of the reference itself: tests/test_reference_golden.py).

The call surface mirrors the reference's native module
(submodules/diff-gaussian-rasterization/rasterize_points.h:18-71): ``rasterize_gaussians``,
``rasterize_gaussians_backward``, ``mark_visible`` -- with numpy arrays instead of torch tensors and
an opaque context object instead of the three byte buffers.  A zero-sized array means "absent",
like the reference's ``torch.Tensor([])`` sentinels (diff_gaussian_rasterization/__init__.py:218-228).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}


def build(force: bool = False) -> None:
    """Compile the oracle's C restatement (gcc, a few seconds)."""
    pairs = [(f"libwg_oracle_{p}.so", "wg_oracle.c") for p in ("f32", "f64")] + [("libwg_knn_oracle.so", "wg_knn_oracle.c")]
    need = force
    for so, src in pairs:
        so, src = os.path.join(_HERE, so), os.path.join(_HERE, src)
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            need = True
    if need:
        subprocess.check_call(["make", "-C", _HERE, "-B", "all"], stdout=subprocess.DEVNULL)


def _lib(precision: str):
    if precision in _LIBS:
        return _LIBS[precision]
    path = os.path.join(_HERE, f"libwg_oracle_{precision}.so")
    if not os.path.exists(path):
        build()
    lib = C.CDLL(path)
    real = C.c_float if precision == "f32" else C.c_double
    p = C.c_void_p
    lib.wgo_forward.restype = C.c_void_p
    lib.wgo_forward.argtypes = [C.c_int, C.c_int, C.c_int, p, C.c_int, C.c_int, p, p, p, p, p, real, p, p, p, p, p,
                                real, real, real, p, C.c_int, p, p]
    lib.wgo_backward.restype = None
    lib.wgo_backward.argtypes = [p, p, p, p, p, p, real, p, p, p, p, p, real, real, real, p, p] + [p] * 9
    lib.wgo_free.argtypes = [p]
    lib.wgo_free.restype = None
    lib.wgo_num_rendered.argtypes = [p]
    lib.wgo_num_rendered.restype = C.c_int
    lib.wgo_mark_visible.argtypes = [C.c_int, p, p, p, p]
    lib.wgo_mark_visible.restype = None
    lib.wgo_get_higher_msb.argtypes = [C.c_uint32]
    lib.wgo_get_higher_msb.restype = C.c_uint32
    _LIBS[precision] = lib
    return lib


def _ptr(a: Optional[np.ndarray]):
    if a is None or a.size == 0:
        return None
    return a.ctypes.data_as(C.c_void_p)


class OracleContext:
    """Owns the intermediate buffers of one forward call (the reference's geom/binning/img buffers)."""

    _FIELDS = {
        "depths": ("real", lambda s: (s.P,)),
        "clamped": (np.uint8, lambda s: (s.P, 3)),
        "radii": (np.int32, lambda s: (s.P,)),
        "means2D": ("real", lambda s: (s.P, 2)),
        "cov3D": ("real", lambda s: (s.P, 6)),
        "conic_opacity": ("real", lambda s: (s.P, 4)),
        "rgb": ("real", lambda s: (s.P, 3)),
        "tiles_touched": (np.uint32, lambda s: (s.P,)),
        "point_offsets": (np.uint32, lambda s: (s.P,)),
        "keys_unsorted": (np.uint64, lambda s: (s.num_rendered,)),
        "keys": (np.uint64, lambda s: (s.num_rendered,)),
        "point_list": (np.uint32, lambda s: (s.num_rendered,)),
        "final_T": ("real", lambda s: (s.H, s.W)),
        "n_contrib": (np.uint32, lambda s: (s.H, s.W)),
        "ranges": (np.uint32, lambda s: (s.tiles, 2)),
        "frag_alpha": ("real", lambda s: (s.H, s.W)),
        "frag_T": ("real", lambda s: (s.H, s.W)),
        "n_evaluated": (np.uint32, lambda s: (s.H, s.W)),
        "n_blended": (np.uint32, lambda s: (s.H, s.W)),
    }

    def __init__(self, lib, handle, precision, P, W, H):
        self._lib, self._h, self.precision = lib, handle, precision
        self.P, self.W, self.H = P, W, H
        self.tiles = ((W + 15) // 16) * ((H + 15) // 16)
        self.num_rendered = lib.wgo_num_rendered(handle) if handle else 0
        self.dtype = np.float32 if precision == "f32" else np.float64

    def get(self, name: str) -> np.ndarray:
        dt, shape_fn = self._FIELDS[name]
        dt = self.dtype if dt == "real" else dt
        out = np.zeros(shape_fn(self), dtype=dt)
        if self._h and out.size:
            fn = getattr(self._lib, f"wgo_get_{name}")
            fn.argtypes = [C.c_void_p, C.c_void_p]
            fn.restype = None
            fn(self._h, out.ctypes.data_as(C.c_void_p))
        return out

    def close(self):
        if self._h:
            self._lib.wgo_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _prep(a, dt):
    if a is None:
        return np.zeros((0,), dtype=dt)
    return np.ascontiguousarray(np.asarray(a), dtype=dt)


def rasterize_gaussians(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                        viewmatrix, projmatrix, tan_fovx, tan_fovy, kernel_size, subpixel_offset, image_height,
                        image_width, sh, degree, campos, prefiltered=False, debug=False, precision="f32"):
    """RasterizeGaussiansCUDA (rasterize_points.cu:35-119) -> (num_rendered, out_color[3,H,W], radii[P], ctx)."""
    lib = _lib(precision)
    dt = np.float32 if precision == "f32" else np.float64
    means3D = _prep(means3D, dt)
    if means3D.ndim != 2 or means3D.shape[1] != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")  # rasterize_points.cu:59-61
    P, H, W = means3D.shape[0], int(image_height), int(image_width)
    arrs = dict(background=_prep(background, dt), colors=_prep(colors, dt), opacity=_prep(opacity, dt),
                scales=_prep(scales, dt), rotations=_prep(rotations, dt), cov3D_precomp=_prep(cov3D_precomp, dt),
                viewmatrix=_prep(viewmatrix, dt), projmatrix=_prep(projmatrix, dt),
                subpixel_offset=_prep(subpixel_offset, dt), sh=_prep(sh, dt), campos=_prep(campos, dt))
    out_color = np.zeros((3, H, W), dtype=dt)
    radii = np.zeros((P,), dtype=np.int32)
    if P == 0:  # rasterize_points.cu:83
        return 0, out_color, radii, OracleContext(lib, None, precision, 0, W, H)
    M = arrs["sh"].shape[1] if arrs["sh"].size else 0  # rasterize_points.cu:85-89
    h = lib.wgo_forward(P, int(degree), M, _ptr(arrs["background"]), W, H, _ptr(means3D), _ptr(arrs["sh"]),
                        _ptr(arrs["colors"]), _ptr(arrs["opacity"]), _ptr(arrs["scales"]), float(scale_modifier),
                        _ptr(arrs["rotations"]), _ptr(arrs["cov3D_precomp"]), _ptr(arrs["viewmatrix"]),
                        _ptr(arrs["projmatrix"]), _ptr(arrs["campos"]), float(tan_fovx), float(tan_fovy),
                        float(kernel_size), _ptr(arrs["subpixel_offset"]), int(bool(prefiltered)),
                        _ptr(out_color), _ptr(radii))
    ctx = OracleContext(lib, h, precision, P, W, H)
    return ctx.num_rendered, out_color, radii, ctx


def rasterize_gaussians_backward(background, means3D, radii, colors, scales, rotations, scale_modifier, cov3D_precomp,
                                 viewmatrix, projmatrix, tan_fovx, tan_fovy, kernel_size, subpixel_offset,
                                 dL_dout_color, sh, degree, campos, ctx: OracleContext, debug=False):
    """RasterizeGaussiansBackwardCUDA (rasterize_points.cu:121-204) ->
    (dL_dmeans2D[P,3], dL_dcolors[P,3], dL_dopacity[P,1], dL_dmeans3D[P,3], dL_dcov3D[P,6], dL_dsh[P,M,3],
     dL_dscales[P,3], dL_drotations[P,4]); also returns dL_dconic[P,2,2] as a 9th element for stage checks."""
    lib, dt = ctx._lib, ctx.dtype
    means3D = _prep(means3D, dt)
    P = means3D.shape[0]
    sh = _prep(sh, dt)
    M = sh.shape[1] if sh.size else 0
    z = lambda *s: np.zeros(s, dtype=dt)
    dmeans3D, dmeans2D, dcolors, dconic = z(P, 3), z(P, 3), z(P, 3), z(P, 2, 2)
    dopacity, dcov3D, dsh, dscales, drot = z(P, 1), z(P, 6), z(P, M, 3), z(P, 3), z(P, 4)
    if P != 0:
        a = dict(background=_prep(background, dt), colors=_prep(colors, dt), scales=_prep(scales, dt),
                 rotations=_prep(rotations, dt), cov3D_precomp=_prep(cov3D_precomp, dt),
                 viewmatrix=_prep(viewmatrix, dt), projmatrix=_prep(projmatrix, dt), campos=_prep(campos, dt),
                 subpixel_offset=_prep(subpixel_offset, dt), dpix=_prep(dL_dout_color, dt))
        lib.wgo_backward(ctx._h, _ptr(a["background"]), _ptr(means3D), _ptr(sh), _ptr(a["colors"]), _ptr(a["scales"]),
                         float(scale_modifier), _ptr(a["rotations"]), _ptr(a["cov3D_precomp"]), _ptr(a["viewmatrix"]),
                         _ptr(a["projmatrix"]), _ptr(a["campos"]), float(tan_fovx), float(tan_fovy),
                         float(kernel_size), _ptr(a["subpixel_offset"]), _ptr(a["dpix"]),
                         dmeans2D.ctypes.data_as(C.c_void_p), dconic.ctypes.data_as(C.c_void_p),
                         dopacity.ctypes.data_as(C.c_void_p), dcolors.ctypes.data_as(C.c_void_p),
                         dmeans3D.ctypes.data_as(C.c_void_p), dcov3D.ctypes.data_as(C.c_void_p),
                         dsh.ctypes.data_as(C.c_void_p) if dsh.size else None,
                         dscales.ctypes.data_as(C.c_void_p), drot.ctypes.data_as(C.c_void_p))
    return dmeans2D, dcolors, dopacity, dmeans3D, dcov3D, dsh, dscales, drot, dconic


def mark_visible(means3D, viewmatrix, projmatrix, precision="f32"):
    """markVisible (rasterize_points.cu:206-225)."""
    lib = _lib(precision)
    dt = np.float32 if precision == "f32" else np.float64
    means3D = _prep(means3D, dt)
    P = means3D.shape[0]
    present = np.zeros((P,), dtype=np.uint8)
    if P:
        v, pm = _prep(viewmatrix, dt), _prep(projmatrix, dt)
        lib.wgo_mark_visible(P, _ptr(means3D), _ptr(v), _ptr(pm), present.ctypes.data_as(C.c_void_p))
    return present.astype(bool)


def get_higher_msb(n: int) -> int:
    return int(_lib("f32").wgo_get_higher_msb(n))


# --------------------------------------------------------------------------------------------------
# Convenience: run forward(+backward) on a wg_scenes-style cloud/camera dict pair.
def run_scene(cloud, cam, sh_degree=3, kernel_size=0.1, bg=None, scale_modifier=1.0, subpixel_offset=None,
              cotangent=None, precision="f32"):
    H, W = cam["height"], cam["width"]
    dt = np.float32 if precision == "f32" else np.float64
    bg = np.zeros(3, dt) if bg is None else bg
    so = np.zeros((H, W, 2), dt) if subpixel_offset is None else subpixel_offset
    e = np.zeros((0,), dt)
    shs = cloud.get("shs", e)
    cols = cloud.get("colors_precomp", e)
    scales = cloud.get("scales", e)
    rots = cloud.get("rotations", e)
    cov = cloud.get("cov3D_precomp", e)
    R, color, radii, ctx = rasterize_gaussians(bg, cloud["means3D"], cols, cloud["opacities"], scales, rots,
                                               scale_modifier, cov, cam["viewmatrix"], cam["projmatrix"],
                                               cam["tanfovx"], cam["tanfovy"], kernel_size, so, H, W, shs, sh_degree,
                                               cam["campos"], False, False, precision)
    out = dict(num_rendered=R, color=color, radii=radii, ctx=ctx, accumulation=1.0 - ctx.get("final_T") if R or True else None)
    if cotangent is not None:
        g = rasterize_gaussians_backward(bg, cloud["means3D"], radii, cols, scales, rots, scale_modifier, cov,
                                         cam["viewmatrix"], cam["projmatrix"], cam["tanfovx"], cam["tanfovy"],
                                         kernel_size, so, cotangent, shs, sh_degree, cam["campos"], ctx)
        names = ["means2D", "colors_precomp", "opacities", "means3D", "cov3Ds_precomp", "sh", "scales", "rotations", "conic"]
        out["grads"] = dict(zip(names, g))
    return out


# --------------------------------------------------------------------------------------------------
# simple_knn._C.distCUDA2 (submodules/simple-knn/spatial.cu:15-26)
def _knn_lib():
    if "knn" not in _LIBS:
        path = os.path.join(_HERE, "libwg_knn_oracle.so")
        if not os.path.exists(path):
            build()
        lib = C.CDLL(path)
        for f in (lib.wgo_knn, lib.wgo_knn_bruteforce):
            f.restype, f.argtypes = None, [C.c_int, C.c_void_p, C.c_void_p]
        _LIBS["knn"] = lib
    return _LIBS["knn"]


def dist_cuda2(points, bruteforce: bool = False) -> np.ndarray:
    """distCUDA2(points[P,3]) -> mean squared distance to the 3 nearest neighbours, float32[P]."""
    pts = np.ascontiguousarray(points, dtype=np.float32)
    P = pts.shape[0]
    out = np.zeros((P,), dtype=np.float32)
    if P:
        fn = _knn_lib().wgo_knn_bruteforce if bruteforce else _knn_lib().wgo_knn
        fn(P, pts.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    return out
