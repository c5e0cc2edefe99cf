"""Fused Gaussian activations + 3-D filter -- opt-in replacement of GaussianModel.get_gaussians's arithmetic
(wildgaussians/method.py:1060-1086).

    from wg_fused_gaussians import activate
    opacities, scales, rotations = activate(raw_opacities, raw_scales, raw_rotations, filter_3D)

with `raw_opacities` [P,1] logits, `raw_scales` [P,3] log-scales, `raw_rotations` [P,4] unnormalised quaternions and the
`filter_3D` [P,1] buffer; the results equal

    rotations = F.normalize(raw_rotations);  s = exp(raw_scales);  scales = sqrt(s^2 + filter_3D^2)
    opacities = sigmoid(raw_opacities) * sqrt(prod(s^2) / prod(s^2 + filter_3D^2))[:, None]

One HIP kernel forward, one backward (include/wg_activations.h, csrc/activations.hip); gradients flow to the three raw
parameters.  No CPU path: float32 tensors on a HIP device.

SURVEY.md 8f N4, the per-Gaussian bookkeeping after the backward pass (method.py:1995-1998, :1470-1477), in place, one kernel,
no host synchronisation (the torch code runs six boolean-mask index operations, each with a nonzero()):

    from wg_fused_gaussians import add_densification_stats
    add_densification_stats(radii, viewspace_points.grad, model.xyz_grad, model.denom, max_radii2D=model.max_radii2D,
                            xyz_gradient_accum_abs=model.xyz_gradient_accum_abs,
                            xyz_gradient_accum_abs_max=model.xyz_gradient_accum_abs_max)
"""
from __future__ import annotations

import ctypes as C

import torch

from diff_gaussian_rasterization import _C as _native

_lib = _native._lib
_vp, _i = C.c_void_p, C.c_int
_lib.wg_activations_forward.restype = _i
_lib.wg_activations_forward.argtypes = [_i] + [_vp] * 8
_lib.wg_activations_backward.restype = _i
_lib.wg_activations_backward.argtypes = [_i] + [_vp] * 11
_lib.wg_densification_stats.restype = _i
_lib.wg_densification_stats.argtypes = [_i] + [_vp] * 8


def _prep(t, cols, name):
    if not (t.is_cuda and t.dtype == torch.float32):
        raise RuntimeError(f"wg_fused_gaussians: {name} must be a float32 tensor on a HIP device (there is no CPU path)")
    if t.numel() % cols:
        raise RuntimeError(f"wg_fused_gaussians: {name} has {t.numel()} elements, not a multiple of {cols}")
    return t.contiguous()


class _Activate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, raw_opacities, raw_scales, raw_rotations, filter_3D):
        o, s, r, f = _prep(raw_opacities, 1, "raw_opacities"), _prep(raw_scales, 3, "raw_scales"), \
            _prep(raw_rotations, 4, "raw_rotations"), _prep(filter_3D, 1, "filter_3D")
        P = o.numel()
        if s.numel() != 3 * P or r.numel() != 4 * P or f.numel() != P:
            raise RuntimeError("wg_fused_gaussians: inconsistent numbers of Gaussians")
        rot, sc, op = torch.empty_like(r), torch.empty_like(s), torch.empty_like(o)
        stream = torch.cuda.current_stream(o.device).cuda_stream
        with torch.cuda.device(o.device):
            _native._check(_lib.wg_activations_forward(P, r.data_ptr(), s.data_ptr(), o.data_ptr(), f.data_ptr(), rot.data_ptr(),
                                                       sc.data_ptr(), op.data_ptr(), stream), "wg_activations_forward")
        ctx.save_for_backward(o, s, r, f)
        return op, sc, rot

    @staticmethod
    def backward(ctx, g_op, g_sc, g_rot):
        o, s, r, f = ctx.saved_tensors
        P = o.numel()
        go, gs, gr = torch.empty_like(o), torch.empty_like(s), torch.empty_like(r)
        ptr = lambda t: t.contiguous().data_ptr() if t is not None else None
        g_op, g_sc, g_rot = [None if t is None else t.contiguous() for t in (g_op, g_sc, g_rot)]
        stream = torch.cuda.current_stream(o.device).cuda_stream
        with torch.cuda.device(o.device):
            _native._check(_lib.wg_activations_backward(P, r.data_ptr(), s.data_ptr(), o.data_ptr(), f.data_ptr(), ptr(g_rot), ptr(g_sc),
                                                        ptr(g_op), gr.data_ptr(), gs.data_ptr(), go.data_ptr(), stream),
                           "wg_activations_backward")
        return go, gs, gr, None


def activate(raw_opacities, raw_scales, raw_rotations, filter_3D):
    """-> (opacities [like raw_opacities], scales [like raw_scales], rotations [like raw_rotations])."""
    return _Activate.apply(raw_opacities, raw_scales, raw_rotations, filter_3D)


def add_densification_stats(radii, viewspace_grad, xyz_grad, denom, max_radii2D=None, xyz_gradient_accum_abs=None,
                            xyz_gradient_accum_abs_max=None):
    """In place, for every Gaussian with radii > 0 (the loop's visibility_filter): xyz_grad += |grad[:, :2]|, denom += 1,
    max_radii2D = max(max_radii2D, radii) and, when the two GOF buffers are given, xyz_gradient_accum_abs += |grad[:, 2]|,
    xyz_gradient_accum_abs_max = max(., |grad[:, 2]|).  `radii` int32 [P], `viewspace_grad` float32 [P, 3], the rest float32 with
    P elements ([P] or [P, 1]), all contiguous on one HIP device.  (A NaN gradient does not propagate into the two max buffers.)"""
    P = radii.numel()
    if not (radii.is_cuda and radii.dtype == torch.int32 and radii.is_contiguous()):
        raise RuntimeError("wg_fused_gaussians: radii must be a contiguous int32 tensor on a HIP device (there is no CPU path)")
    if (xyz_gradient_accum_abs is None) != (xyz_gradient_accum_abs_max is None):
        raise RuntimeError("wg_fused_gaussians: pass both GOF buffers or neither")
    g = _prep(viewspace_grad, 3, "viewspace_grad")
    if g.numel() != 3 * P:
        raise RuntimeError("wg_fused_gaussians: viewspace_grad must be [P, 3]")
    bufs = []
    for name, t in (("xyz_grad", xyz_grad), ("xyz_gradient_accum_abs", xyz_gradient_accum_abs),
                    ("xyz_gradient_accum_abs_max", xyz_gradient_accum_abs_max), ("denom", denom), ("max_radii2D", max_radii2D)):
        if t is None:
            bufs.append(None)
            continue
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.numel() == P):
            raise RuntimeError(f"wg_fused_gaussians: {name} must be a contiguous float32 tensor of {P} elements on the HIP device")
        bufs.append(t.data_ptr())
    stream = torch.cuda.current_stream(radii.device).cuda_stream
    with torch.cuda.device(radii.device):
        _native._check(_lib.wg_densification_stats(P, radii.data_ptr(), g.data_ptr(), bufs[0], bufs[1], bufs[2], bufs[3], bufs[4], stream),
                       "wg_densification_stats")
