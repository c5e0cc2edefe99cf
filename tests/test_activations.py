"""fused activations + 3-D filter (include/wg_activations.h, wg_fused_gaussians) against a plain PyTorch
float32 restatement of GaussianModel.get_gaussians (wildgaussians/method.py:1060-1086; activations method.py:923-927)."""
import os
import sys

import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wild-gaussians_amd"))


def ref_get_gaussians(raw_opacities, raw_scales, raw_rotations, filter_3D):
    rotations = F.normalize(raw_rotations)
    raw_s = torch.exp(raw_scales)
    opacities = torch.sigmoid(raw_opacities)
    scales = (torch.square(raw_s) + torch.square(filter_3D)).sqrt()
    scales_square = torch.square(raw_s)
    det1 = scales_square.prod(dim=1)
    det2 = (scales_square + torch.square(filter_3D)).prod(dim=1)
    coef = torch.sqrt(det1 / det2)
    return opacities * coef[..., None], scales, rotations


def test_restatement_follows_method_py():
    ref = "/root/reference/wildgaussians/method.py"
    if not os.path.isfile(ref):
        pytest.skip("reference checkout not present")
    src = open(ref).read()
    body = src[src.index("    def get_gaussians(self):"):src.index("    def _resize_parameter(self")]
    for frag in ("self.rotation_activation(self.rotations)", "self.scaling_activation(self.scales)", "self.opacity_activation(self.opacities)",
                 "(torch.square(raw_scales) + torch.square(self.filter_3D)).sqrt_()", "coef = torch.sqrt(det1 / det2)",
                 "opacities = opacities * coef[..., None]"):
        assert frag in body, frag
    assert "scaling_activation = staticmethod(torch.exp)" in src and "opacity_activation = staticmethod(torch.sigmoid)" in src
    assert "rotation_activation = staticmethod(torch.nn.functional.normalize)" in src


@pytest.mark.gpu
@pytest.mark.parametrize("P", [1, 255, 4097, 100000])
def test_fused_activations_match_the_reference_formulas(P):
    from wg_fused_gaussians import activate
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(P)
    mk = lambda *shape, scale=1.0, shift=0.0: (torch.randn(*shape, generator=g) * scale + shift).to(dev)
    ro, rs, rr = mk(P, 1, scale=2.0), mk(P, 3, scale=0.7, shift=-4.0), mk(P, 4)
    f3 = (torch.rand(P, 1, generator=g) * 0.03).to(dev)
    if P > 300:
        f3[:50] = 0.0           # no filter: coef == 1
        rr[50:60] = 0.0         # zero quaternion: F.normalize divides by eps
    a = [t.clone().requires_grad_(True) for t in (ro, rs, rr)]
    b = [t.clone().requires_grad_(True) for t in (ro, rs, rr)]
    out = activate(a[0], a[1], a[2], f3)
    ref = ref_get_gaussians(b[0], b[1], b[2], f3)
    for x, y, name in zip(out, ref, ("opacities", "scales", "rotations")):
        assert x.shape == y.shape, name
        assert ((x - y).abs() <= 1e-6 + 2e-6 * y.abs()).all(), (name, (x - y).abs().max().item())
    w = [torch.randn(t.shape, generator=g).to(dev) for t in ref]
    sum((x * k).sum() for x, k in zip(out, w)).backward()
    sum((y * k).sum() for y, k in zip(ref, w)).backward()
    for p, q, name in zip(a, b, ("raw_opacities", "raw_scales", "raw_rotations")):
        err = (p.grad - q.grad).abs().max().item() / (q.grad.abs().max().item() + 1e-12)
        assert err <= 2e-5, (name, err)


@pytest.mark.gpu
def test_fused_activations_refuse_host_tensors():
    from wg_fused_gaussians import activate
    with pytest.raises(RuntimeError):
        activate(torch.zeros(4, 1), torch.zeros(4, 3), torch.zeros(4, 4), torch.zeros(4, 1))


def test_c_abi_exports_the_activation_entry_points():
    import ctypes as C
    import re
    lib = C.CDLL(os.path.join(ROOT, "wild-gaussians_amd", "diff_gaussian_rasterization", "libwg_rasterizer.so"))
    names = set(re.findall(r"\b(wg_activations_\w+)\s*\(", open(os.path.join(ROOT, "include", "wg_activations.h")).read()))
    assert names == {"wg_activations_forward", "wg_activations_backward"}
    for n in names:
        assert hasattr(lib, n)
    lib.wg_activations_forward.restype = C.c_int
    lib.wg_activations_forward.argtypes = [C.c_int] + [C.c_void_p] * 8
    assert lib.wg_activations_forward(-1, *([None] * 8)) == -1
    assert lib.wg_activations_forward(0, *([None] * 8)) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("colors", ["precomp", "sh_toned"])
def test_raw_parameter_mode_of_the_operator_equals_activations_then_plain_call(colors):
    """`filter_3D=` (wg_raw_gaussians; SURVEY 8f N3 as worded: the activations and the 3-D filter INSIDE the preprocess kernels): the
    operator takes the raw opacities / scales / rotations and runs get_gaussians() (method.py:1060-1086) itself, forward and backward.
    Against fused activations (wg_fused_gaussians.activate: the same device functions, the same compiler flags) followed by the plain
    call: image, radii and accumulation bit-identical; every gradient -- those of the raw parameters included -- equal to rounding; and
    against the PyTorch restatement of get_gaussians within float32 tolerance.  With `shs` + `sh_mul` / `sh_offset` on top, the whole
    step in front of the operator (activations, filter, toning, SH evaluation) is in-kernel."""
    import numpy as np
    import wg_scenes as S
    import wg_fused_gaussians as FG
    from diff_gaussian_rasterization import GaussianRasterizer
    from tests.wg_testlib import make_settings, to_dev
    P, W, H = 80_000, 640, 400
    sh = colors == "sh_toned"
    cloud = S.make_cloud(P, W, H, sh_degree=3 if sh else None, seed=9, scale_mult=2.0)
    cam = S.make_camera(W, H, yaw_deg=2.0)
    rs = make_settings(cam, 3 if sh else 0)
    g = torch.Generator().manual_seed(4)
    filt = (0.3 * torch.rand(P, 1, generator=g) * torch.from_numpy(cloud["scales"]).mean(dim=1, keepdim=True)).cuda()
    # raw parameters whose activations reproduce the recipe's cloud (before the filter)
    raw = dict(opacities=torch.special.logit(torch.from_numpy(cloud["opacities"]).clamp(1e-4, 1 - 1e-4)).cuda(),
               scales=torch.log(torch.from_numpy(cloud["scales"])).cuda(),
               rotations=(torch.from_numpy(cloud["rotations"]) * (0.5 + torch.rand(P, 1, generator=g))).cuda())
    cot = to_dev(S.make_cotangent(W, H, seed=6))
    tone = {}
    if sh:
        tone = dict(sh_mul=(0.8 + 0.4 * torch.rand(P, 3, generator=g)).cuda(), sh_offset=(0.2 * torch.rand(P, 3, generator=g)).cuda(),
                    sh_pre_clamp_max=1.0, sh_post_clamp_max=1.0)

    def run(mode):
        t = {k: v.clone().requires_grad_(True) for k, v in raw.items()}
        m3 = to_dev(cloud["means3D"]).requires_grad_(True)
        col = to_dev(cloud["shs"] if sh else cloud["colors_precomp"]).requires_grad_(True)
        m2d = torch.zeros((P, 3), device="cuda", requires_grad=True)
        tn = {k: (v.clone().requires_grad_(True) if torch.is_tensor(v) else v) for k, v in tone.items()}
        kw = dict(shs=col, **tn) if sh else dict(colors_precomp=col)
        rast = GaussianRasterizer(rs)
        if mode == "raw":
            img, radii, acc = rast(means3D=m3, means2D=m2d, opacities=t["opacities"], scales=t["scales"], rotations=t["rotations"], filter_3D=filt, **kw)
        else:
            act = FG.activate if mode == "fused" else ref_get_gaussians
            o, s_, r = act(t["opacities"], t["scales"], t["rotations"], filt)
            img, radii, acc = rast(means3D=m3, means2D=m2d, opacities=o, scales=s_, rotations=r, **kw)
        (img * cot).sum().backward()
        out = dict(img=img, radii=radii, acc=acc, g_m3=m3.grad, g_col=col.grad, g_m2d=m2d.grad, **{"g_" + k: v.grad for k, v in t.items()})
        out.update({"g_" + k: v.grad for k, v in tn.items() if torch.is_tensor(v)})
        return {k: v.detach().cpu().numpy() for k, v in out.items()}
    a, b, c = run("raw"), run("fused"), run("torch")
    for k in ("img", "radii", "acc"):
        assert np.array_equal(a[k], b[k]), k
    assert int((a["radii"] > 0).sum()) > P // 2 and a["img"].any()
    for k in a:
        if k.startswith("g_"):
            scale = float(np.abs(b[k]).max())
            assert scale > 0 and float(np.abs(a[k] - b[k]).max()) <= 1e-4 * scale, (k, float(np.abs(a[k] - b[k]).max()), scale)
            assert float(np.abs(a[k] - c[k]).max()) <= 2e-3 * float(np.abs(c[k]).max()), (k, "vs torch restatement")
    assert (a["radii"] != c["radii"]).sum() <= max(2, P // 50_000) and float(np.quantile(np.abs(a["img"] - c["img"]), 0.9999)) <= 1e-5
