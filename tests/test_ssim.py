"""fused SSIM (include/wg_ssim.h, wg_fused_ssim) against a plain PyTorch float32 restatement of the
reference's ssim() (wildgaussians/method.py:644-673) -- a floating-point kernel, so the torch reference is the oracle."""
import math
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wild-gaussians_amd"))


def ref_ssim(img1, img2, window_size=11, size_average=True):
    """method.py:644-673, restated."""
    sigma = 1.5
    channel = img1.size(-3)
    gauss = torch.Tensor([math.exp(-(x - window_size // 2) ** 2 / float(2 * sigma ** 2)) for x in range(window_size)])
    w1 = (gauss / gauss.sum()).unsqueeze(1)
    window = w1.mm(w1.t()).float().unsqueeze(0).unsqueeze(0).expand(channel, 1, window_size, window_size).contiguous()
    window = window.to(img1.device).type_as(img1)
    conv = lambda t: F.conv2d(t, window, padding=window_size // 2, groups=channel)
    mu1, mu2 = conv(img1), conv(img2)
    mu1_sq, mu2_sq, mu1_mu2 = mu1.pow(2), mu2.pow(2), mu1 * mu2
    s1, s2, s12 = conv(img1 * img1) - mu1_sq, conv(img2 * img2) - mu2_sq, conv(img1 * img2) - mu1_mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    m = ((2 * mu1_mu2 + C1) * (2 * s12 + C2)) / ((mu1_sq + mu2_sq + C1) * (s1 + s2 + C2))
    return m.mean() if size_average else m.mean(-3)


def test_reference_restatement_matches_method_py():
    """Where the reference checkout exists, the restatement above is checked against the real function (CPU)."""
    ref = "/root/reference"
    if not os.path.isdir(os.path.join(ref, "wildgaussians")):
        pytest.skip("reference checkout not present")
    src = open(os.path.join(ref, "wildgaussians", "method.py")).read()
    a = src.index("def ssim(img1, img2, window_size=11, size_average=True):")
    b = src.index("\n\n\n", a)
    ns = {"torch": torch, "F": F, "math": math}
    exec(src[a:b], ns)  # the function body only: method.py itself needs packages this image lacks
    g = torch.Generator().manual_seed(0)
    x, y = torch.rand(3, 37, 53, generator=g), torch.rand(3, 37, 53, generator=g)
    assert torch.equal(ns["ssim"](x[None], y[None]), ref_ssim(x[None], y[None]))
    assert torch.equal(ns["ssim"](x[None], y[None], size_average=False), ref_ssim(x[None], y[None], size_average=False))


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(3, 64, 96), (3, 33, 47), (1, 16, 16), (3, 7, 5), (3, 200, 300)])
@pytest.mark.parametrize("size_average", [True, False])
def test_fused_ssim_matches_the_reference_formula(shape, size_average):
    from wg_fused_ssim import ssim
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.rand(shape, generator=g).to(dev).requires_grad_(True)
    y = (torch.rand(shape, generator=g) * 0.5 + 0.25 * x.detach().cpu()).to(dev)
    xr = x.detach().clone().requires_grad_(True)
    out = ssim(x, y, size_average=size_average)
    ref = ref_ssim(xr[None], y[None], size_average=size_average)
    ref = ref if size_average else ref[0]
    assert out.shape == ref.shape
    assert (out - ref).abs().max().item() <= 2e-5
    wgt = torch.rand(out.shape, generator=g).to(dev) if not size_average else torch.tensor(1.0, device=dev)
    (out * wgt).sum().backward()
    (ref * wgt).sum().backward()
    err = (x.grad - xr.grad).abs().max().item() / (xr.grad.abs().max().item() + 1e-12)
    assert err <= 1e-4, err


@pytest.mark.gpu
def test_fused_ssim_full_frame_properties():
    """1600x1200 (BASELINE config 3's frame): ssim(x, x) == 1 everywhere, symmetric in its arguments, bounded by 1."""
    from wg_fused_ssim import ssim, ssim_map
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(4)
    x, y = torch.rand(3, 1200, 1600, generator=g).to(dev), torch.rand(3, 1200, 1600, generator=g).to(dev)
    assert (ssim_map(x, x) - 1.0).abs().max().item() <= 1e-5
    a, b = ssim_map(x, y), ssim_map(y, x)
    assert (a - b).abs().max().item() <= 1e-5
    assert a.max().item() <= 1.0 + 1e-6
    with pytest.raises(RuntimeError):
        ssim(x.cpu(), y.cpu())  # no CPU path


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(3, 64, 96), (3, 33, 47), (3, 120, 200)])
@pytest.mark.parametrize("mode", ["two_images", "one_image", "with_mult"])
def test_fused_l1_ssim_loss_matches_the_reference_statements(shape, mode):
    """wg_fused_ssim.l1_ssim_loss against the loss statements of the reference's train step (method.py:1948-1965) written with the
    torch restatement of ssim(): value, the two logged parts, and both image gradients."""
    from wg_fused_ssim import l1_ssim_loss
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(sum(shape) + len(mode))
    lam = 0.2
    gt = torch.rand(shape, generator=g).to(dev)
    toned = (gt.cpu() * 0.6 + 0.4 * torch.rand(shape, generator=g)).to(dev).requires_grad_(True)
    raw = (gt.cpu() * 0.5 + 0.5 * torch.rand(shape, generator=g)).to(dev).requires_grad_(True)
    mult = (torch.rand((1,) + shape[1:], generator=g) * 2.0).to(dev) if mode == "with_mult" else None
    if mode == "one_image":
        raw = toned
    t2, r2 = toned.detach().clone().requires_grad_(True), None
    r2 = t2 if mode == "one_image" else raw.detach().clone().requires_grad_(True)
    loss, l1m, ssm = l1_ssim_loss(toned, raw, gt, lam, loss_mult=mult, return_parts=True)
    lm = 1.0 if mult is None else mult
    Ll1 = F.l1_loss(t2, gt, reduction="none")
    ssim_value = ref_ssim(r2[None], gt[None], size_average=False)[0]
    ref = (1.0 - lam) * (Ll1 * lm).mean() + lam * ((1.0 - ssim_value) * lm).mean()
    assert abs(loss.item() - ref.item()) <= 2e-6 * max(1.0, abs(ref.item()))
    assert abs(l1m.item() - (Ll1 * lm).mean().item()) <= 2e-6 and not l1m.requires_grad
    assert abs(ssm.item() - (1.0 - ((1.0 - ssim_value) * lm).mean().item())) <= 2e-5
    (loss * 3.0).backward()
    (ref * 3.0).backward()
    for a, b in ((toned, t2),) + (() if mode == "one_image" else ((raw, r2),)):
        err = (a.grad - b.grad).abs().max().item() / (b.grad.abs().max().item() + 1e-12)
        assert err <= 1e-4, (mode, err)
    # bit-reproducible value (fixed summation order) and the constants are refused as differentiable inputs
    again = l1_ssim_loss(toned, raw, gt, lam, loss_mult=mult)
    assert again.item() == loss.item()
    with pytest.raises(RuntimeError):
        l1_ssim_loss(toned, raw, gt.clone().requires_grad_(True), lam)


@pytest.mark.gpu
def test_fused_ssim_refuses_a_differentiable_second_image_and_runs_without_gradients():
    from wg_fused_ssim import ssim
    dev = torch.device("cuda", 0)
    x, y = torch.rand(3, 40, 50, device=dev), torch.rand(3, 40, 50, device=dev)
    with pytest.raises(RuntimeError):
        ssim(x, y.clone().requires_grad_(True))
    with torch.no_grad():
        v = ssim(x.clone().requires_grad_(True), y)
    assert not v.requires_grad and 0.0 < v.item() < 1.0


def test_c_abi_exports_the_ssim_entry_points():
    import ctypes as C
    lib = C.CDLL(os.path.join(ROOT, "wild-gaussians_amd", "diff_gaussian_rasterization", "libwg_rasterizer.so"))
    hdr = open(os.path.join(ROOT, "include", "wg_ssim.h")).read()
    import re
    names = set(re.findall(r"\b(wg_(?:l1_)?ssim_\w+)\s*\(", hdr))
    assert names == {"wg_ssim_forward", "wg_ssim_backward", "wg_l1_ssim_loss_forward", "wg_l1_ssim_loss_backward", "wg_l1_ssim_loss_scratch_floats"}
    for n in names:
        assert hasattr(lib, n)
    lib.wg_ssim_forward.restype = C.c_int
    lib.wg_ssim_forward.argtypes = [C.c_int] * 3 + [C.c_void_p] * 7
    assert lib.wg_ssim_forward(3, 0, 4, None, None, None, None, None, None, None) == -1  # rejected before any device work
    lib.wg_l1_ssim_loss_scratch_floats.restype = C.c_size_t
    assert lib.wg_l1_ssim_loss_scratch_floats(3, 1200, 1600) == 2 * 50 * 75 * 3 and lib.wg_l1_ssim_loss_scratch_floats(0, 4, 4) == 0
