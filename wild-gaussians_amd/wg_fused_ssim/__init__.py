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
        need = img1.requires_grad
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
