"""Fused SSIM -- opt-in replacement of the reference's `ssim()` (wildgaussians/method.py:644-673).

    from wg_fused_ssim import ssim          # same signature and results as method.py's ssim
    value = ssim(image, gt_image)                        # scalar (size_average=True)
    smap = ssim(image, gt_image, size_average=False)     # [H, W]: mean over channels, as method.py:1949 uses it

One HIP kernel computes the five 11x11 Gaussian-window statistics, the SSIM map and the three partial-derivative maps the
backward needs; one more kernel is the whole backward (include/wg_ssim.h, csrc/ssim.hip).  Gradients flow to `img1` only:
`img2` is the ground truth in the reference's use.  There is no CPU fallback: tensors must be float32 on a HIP device.
"""
from __future__ import annotations

import ctypes as C

import torch

from diff_gaussian_rasterization import _C as _native

_lib = _native._lib
_vp, _i = C.c_void_p, C.c_int
_lib.wg_ssim_forward.restype = _i
_lib.wg_ssim_forward.argtypes = [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]
_lib.wg_ssim_backward.restype = _i
_lib.wg_ssim_backward.argtypes = [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]
_f = C.c_float
_lib.wg_l1_ssim_loss_scratch_floats.restype = C.c_size_t
_lib.wg_l1_ssim_loss_scratch_floats.argtypes = [_i, _i, _i]
_lib.wg_l1_ssim_loss_forward.restype = _i
_lib.wg_l1_ssim_loss_forward.argtypes = [_i, _i, _i, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _vp]
_lib.wg_l1_ssim_loss_backward.restype = _i
_lib.wg_l1_ssim_loss_backward.argtypes = [_i, _i, _i, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp]


def _check(img):
    if not (img.is_cuda and img.dtype == torch.float32):
        raise RuntimeError("wg_fused_ssim: float32 tensors on a HIP device are required (there is no CPU path)")


class _SSIMMap(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img1, img2):
        _check(img1)
        _check(img2)
        if img1.shape != img2.shape or img1.dim() != 3:
            raise RuntimeError("wg_fused_ssim: two [C, H, W] images of the same shape are expected")
        a, b = img1.contiguous(), img2.contiguous()
        Cn, H, W = a.shape
        out = torch.empty_like(a)
        # Gradients flow to img1 only (img2 is the ground truth in the reference's use, method.py:1949); asking for img2's is an
        # error here rather than a silent None -- the reference's torch ssim() would differentiate both.
        if img2.requires_grad:
            raise RuntimeError("wg_fused_ssim: img2 is treated as a constant (the ground truth); detach it, or swap the arguments "
                               "(SSIM is symmetric) to differentiate with respect to it")
        need = ctx.needs_input_grad[0]
        ctx.have_maps = need
        d = torch.empty((3, Cn, H, W), device=a.device, dtype=torch.float32) if need else None
        stream = torch.cuda.current_stream(a.device).cuda_stream
        with torch.cuda.device(a.device):
            _native._check(_lib.wg_ssim_forward(Cn, H, W, a.data_ptr(), b.data_ptr(), out.data_ptr(),
                                                d[0].data_ptr() if need else None, d[1].data_ptr() if need else None,
                                                d[2].data_ptr() if need else None, stream), "wg_ssim_forward")
        if need:
            ctx.save_for_backward(a, b, d)
        return out

    @staticmethod
    def backward(ctx, grad_map):
        if not ctx.have_maps:
            return None, None
        a, b, d = ctx.saved_tensors
        Cn, H, W = a.shape
        g = grad_map.contiguous()
        out = torch.empty_like(a)
        stream = torch.cuda.current_stream(a.device).cuda_stream
        with torch.cuda.device(a.device):
            _native._check(_lib.wg_ssim_backward(Cn, H, W, a.data_ptr(), b.data_ptr(), g.data_ptr(), d[0].data_ptr(), d[1].data_ptr(),
                                                 d[2].data_ptr(), out.data_ptr(), stream), "wg_ssim_backward")
        return out, None


def ssim_map(img1: torch.Tensor, img2: torch.Tensor) -> torch.Tensor:
    """Per-channel SSIM map [C, H, W]."""
    return _SSIMMap.apply(img1, img2)


def ssim(img1: torch.Tensor, img2: torch.Tensor, window_size: int = 11, size_average: bool = True) -> torch.Tensor:
    """method.py:644-673.  Only the reference's window (11, sigma 1.5) is built in."""
    if window_size != 11:
        raise ValueError("wg_fused_ssim.ssim: only window_size=11 (the reference's default) is implemented")
    lead = img1.shape[:-3]
    if lead:  # batched input [..., C, H, W]: one launch per image
        maps = torch.stack([_SSIMMap.apply(x, y) for x, y in zip(img1.reshape(-1, *img1.shape[-3:]), img2.reshape(-1, *img2.shape[-3:]))])
        maps = maps.reshape(*lead, *img1.shape[-3:])
    else:
        maps = _SSIMMap.apply(img1, img2)
    return maps.mean() if size_average else maps.mean(-3)


class _L1SSIMLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img_l1, img_ssim, gt, lambda_dssim, loss_mult):
        for t in (img_l1, img_ssim, gt):
            _check(t)
        if not (img_l1.shape == img_ssim.shape == gt.shape) or gt.dim() != 3:
            raise RuntimeError("wg_fused_ssim.l1_ssim_loss: three [C, H, W] images of the same shape are expected")
        if gt.requires_grad or (loss_mult is not None and loss_mult.requires_grad):
            raise RuntimeError("wg_fused_ssim.l1_ssim_loss: the ground truth and loss_mult are constants; detach them")
        same = img_l1 is img_ssim
        a = img_l1.contiguous()
        b = a if same else img_ssim.contiguous()
        g = gt.contiguous()
        Cn, H, W = g.shape
        m = None
        if loss_mult is not None:
            _check(loss_mult)
            m = loss_mult.expand(1, H, W).contiguous() if loss_mult.dim() == 3 else loss_mult.reshape(H, W).contiguous()
        need = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        dev = g.device
        d = torch.empty((3, Cn, H, W), device=dev, dtype=torch.float32) if need else None
        scratch = torch.empty((_lib.wg_l1_ssim_loss_scratch_floats(Cn, H, W),), device=dev, dtype=torch.float32)
        out = torch.empty((3,), device=dev, dtype=torch.float32)
        stream = torch.cuda.current_stream(dev).cuda_stream
        with torch.cuda.device(dev):
            _native._check(_lib.wg_l1_ssim_loss_forward(Cn, H, W, a.data_ptr(), b.data_ptr(), g.data_ptr(), None if m is None else m.data_ptr(),
                                                        float(lambda_dssim), scratch.data_ptr(), out.data_ptr(),
                                                        d[0].data_ptr() if need else None, d[1].data_ptr() if need else None,
                                                        d[2].data_ptr() if need else None, stream), "wg_l1_ssim_loss_forward")
        ctx.lam, ctx.same, ctx.have_maps = float(lambda_dssim), same, need
        if need:
            ctx.save_for_backward(a, b, g, d, m if m is not None else torch.empty(0, device=dev))
        loss, parts = out[0], out[1:]
        ctx.mark_non_differentiable(parts)
        return loss, parts

    @staticmethod
    def backward(ctx, grad_loss, _grad_parts):
        if not ctx.have_maps:
            return None, None, None, None, None
        a, b, g, d, m = ctx.saved_tensors
        Cn, H, W = g.shape
        gl = grad_loss.contiguous().reshape(1).float()
        ga = torch.empty_like(a)
        gb = ga if ctx.same else torch.empty_like(b)
        stream = torch.cuda.current_stream(g.device).cuda_stream
        with torch.cuda.device(g.device):
            _native._check(_lib.wg_l1_ssim_loss_backward(Cn, H, W, a.data_ptr(), b.data_ptr(), g.data_ptr(), m.data_ptr() if m.numel() else None,
                                                         ctx.lam, gl.data_ptr(), d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(),
                                                         ga.data_ptr(), gb.data_ptr(), stream), "wg_l1_ssim_loss_backward")
        if ctx.same:   # autograd adds the two returned gradients of the one tensor: hand the sum over once
            return ga, None, None, None, None
        return ga, gb, None, None, None


def l1_ssim_loss(img_l1: torch.Tensor, img_ssim: torch.Tensor, gt: torch.Tensor, lambda_dssim: float = 0.2, loss_mult=None,
                 return_parts: bool = False):
    """The image loss of the reference's train step (wildgaussians/method.py:1948-1965) in two kernel launches each way:

        (1 - lambda_dssim) * (|img_l1 - gt| * loss_mult).mean() + lambda_dssim * ((1 - ssim(img_ssim, gt, size_average=False)) * loss_mult).mean()

    img_l1 is the appearance-toned render, img_ssim the raw one (pass the same tensor twice when there is only one); gt and the
    optional per-pixel loss_mult ([H, W] or [1, H, W]) are constants.  return_parts=True also returns the detached
    (l1_mean, ssim_mean) the reference logs (method.py:1970-1972)."""
    loss, parts = _L1SSIMLoss.apply(img_l1, img_ssim, gt, lambda_dssim, loss_mult)
    return (loss, parts[0], parts[1]) if return_parts else loss
