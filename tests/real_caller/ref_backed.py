"""A `diff_gaussian_rasterization.GaussianRasterizer` stand-in whose kernels are the REFERENCE'S OWN (oracle/_ref: its CUDA sources built
for gfx950 by oracle/ref_hip/Makefile), for trajectory-level parity: the reference's real `train_iteration` trained once on this
repo's operator and once on this stand-in, from one seed (tests/test_real_caller.py).

TEST INFRASTRUCTURE ONLY -- never imported by the product.  It mirrors the reference's Python wrapper
(submodules/diff-gaussian-rasterization/diff_gaussian_rasterization/__init__.py:44-173): same inputs, same gradient order, the
accumulation read from the image state.  oracle/ref_hip/driver.cpp keeps ONE set of scratch buffers (the backward pass is that of
the last forward pass), while WildGaussians rasterizes twice before it differentiates (method.py:1573-1611) -- so the backward pass
here re-runs its own forward pass first.  Everything runs on the null stream, which is torch's default stream on ROCm.
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn

from oracle.ref_hip import ref_hip


def _dp(t):
    return None if t is None or t.numel() == 0 else C.c_void_p(t.data_ptr())


def _f(t):
    return t if t is None or t.numel() == 0 else t.detach().float().contiguous()


def _cam(rs):
    """The settings' tensors as contiguous float32 -- kept alive by the caller for the duration of the native call (a `.T` view, as
    the caller's viewmatrix is, becomes a temporary whose block the next temporary would reuse)."""
    return dict(bg=_f(rs.bg), view=_f(rs.viewmatrix), proj=_f(rs.projmatrix), campos=_f(rs.campos), so=_f(rs.subpixel_offset))


def _forward(lib, a, rs, want_state=True):
    c = _cam(rs)
    P = a["means3D"].shape[0]
    H, W = int(rs.image_height), int(rs.image_width)
    dev = a["means3D"].device
    M = a["sh"].shape[1] if a["sh"].numel() else 0
    color = torch.zeros((3, H, W), device=dev)
    radii = torch.zeros((P,), dtype=torch.int32, device=dev)
    final_T = torch.zeros((H * W,), device=dev) if want_state else None
    n_contrib = torch.zeros((H * W,), dtype=torch.int32, device=dev) if want_state else None
    R = lib.refhip_forward(P, int(rs.sh_degree), M, _dp(c["bg"]), W, H, _dp(a["means3D"]), _dp(a["sh"]), _dp(a["colors_precomp"]),
                           _dp(a["opacities"]), _dp(a["scales"]), float(rs.scale_modifier), _dp(a["rotations"]), _dp(a["cov3Ds_precomp"]),
                           _dp(c["view"]), _dp(c["proj"]), _dp(c["campos"]), float(rs.tanfovx), float(rs.tanfovy),
                           float(rs.kernel_size), _dp(c["so"]), 0, _dp(color), _dp(radii), _dp(final_T), _dp(n_contrib), 0)
    if R < 0:
        raise MemoryError("reference build: scratch allocation failed")
    return R, color, radii, final_T


def _backward(lib, a, rs, radii, grad_out):
    c = _cam(rs)
    grad_out = _f(grad_out)
    P = a["means3D"].shape[0]
    H, W = int(rs.image_height), int(rs.image_width)
    dev = a["means3D"].device
    M = a["sh"].shape[1] if a["sh"].numel() else 0
    z = lambda *s: torch.zeros(s, device=dev)  # noqa: E731
    g = dict(means2D=z(P, 3), conic=z(P, 2, 2), opacities=z(P, 1), colors_precomp=z(P, 3), means3D=z(P, 3), cov3Ds_precomp=z(P, 6),
             sh=z(P, max(M, 1), 3), scales=z(P, 3), rotations=z(P, 4))
    lib.refhip_backward(P, int(rs.sh_degree), M, _dp(c["bg"]), W, H, _dp(a["means3D"]), _dp(a["sh"]), _dp(a["colors_precomp"]), _dp(a["scales"]),
                        float(rs.scale_modifier), _dp(a["rotations"]), _dp(a["cov3Ds_precomp"]), _dp(c["view"]), _dp(c["proj"]),
                        _dp(c["campos"]), float(rs.tanfovx), float(rs.tanfovy), float(rs.kernel_size), _dp(c["so"]), _dp(radii),
                        _dp(grad_out), _dp(g["means2D"]), _dp(g["conic"]), _dp(g["opacities"]), _dp(g["colors_precomp"]), _dp(g["means3D"]),
                        _dp(g["cov3Ds_precomp"]), _dp(g["sh"]), _dp(g["scales"]), _dp(g["rotations"]), 0)
    if M == 0:
        g["sh"] = torch.zeros((P, 0, 3), device=dev)
    return g


def make(variant: str = "nofma"):
    """-> a GaussianRasterizer class backed by oracle/_ref's `variant` build."""
    lib = ref_hip._lib(variant)

    class _Fn(torch.autograd.Function):
        @staticmethod
        def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, rs):
            a = dict(means3D=_f(means3D), sh=_f(sh), colors_precomp=_f(colors_precomp), opacities=_f(opacities), scales=_f(scales),
                     rotations=_f(rotations), cov3Ds_precomp=_f(cov3Ds_precomp))
            R, color, radii, final_T = _forward(lib, a, rs)
            ctx.rs, ctx.a = rs, a
            ctx.set_materialize_grads(False)
            ctx.save_for_backward(radii)
            acc = (1.0 - final_T).view(rs.image_height, rs.image_width) if rs.return_accumulation else None
            ctx.mark_non_differentiable(radii)
            return color, radii, acc

        @staticmethod
        def backward(ctx, grad_out, _r, _a):
            rs, a = ctx.rs, ctx.a
            if grad_out is None:
                grad_out = torch.zeros((3, rs.image_height, rs.image_width), device=a["means3D"].device)
            _R, _c, radii, _t = _forward(lib, a, rs, want_state=False)   # the driver's scratch is that of the LAST forward pass
            g = _backward(lib, a, rs, radii, grad_out)
            e = lambda k: g[k] if a[k].numel() else None  # noqa: E731
            return (g["means3D"], g["means2D"], e("sh"), e("colors_precomp"), g["opacities"], e("scales"), e("rotations"),
                    e("cov3Ds_precomp"), None)

    class RefBackedRasterizer(nn.Module):
        def __init__(self, raster_settings):
            super().__init__()
            self.raster_settings = raster_settings

        def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None):
            e = torch.Tensor([])
            return _Fn.apply(means3D, means2D, e if shs is None else shs, e if colors_precomp is None else colors_precomp, opacities,
                             e if scales is None else scales, e if rotations is None else rotations,
                             e if cov3D_precomp is None else cov3D_precomp, self.raster_settings)

    return RefBackedRasterizer


def replay(call_kwargs, rs, grad_out, variant: str = "nofma"):
    """One recorded rasterizer call (harness.RasterizerTap) through the reference build, forward + backward: -> (color, radii, grads)."""
    lib = ref_hip._lib(variant)
    e = torch.Tensor([])
    get = lambda k: _f(call_kwargs[k]) if call_kwargs.get(k) is not None else e  # noqa: E731
    a = dict(means3D=get("means3D"), sh=get("shs"), colors_precomp=get("colors_precomp"), opacities=get("opacities"), scales=get("scales"),
             rotations=get("rotations"), cov3Ds_precomp=get("cov3D_precomp"))
    _R, color, radii, _t = _forward(lib, a, rs)
    g = _backward(lib, a, rs, radii, grad_out)
    torch.cuda.synchronize()
    return color, radii, g
