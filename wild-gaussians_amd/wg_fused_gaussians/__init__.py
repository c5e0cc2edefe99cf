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
