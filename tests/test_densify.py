"""fused densification statistics (include/wg_densify.h, wg_fused_gaussians.add_densification_stats) against a
plain PyTorch float32 restatement of wildgaussians/method.py:1995-1998 and GaussianModel.add_densification_stats (:1470-1477)."""
import ctypes as C
import os
import re
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wild-gaussians_amd"))


def ref_stats(radii, grad, xyz_grad, denom, max_radii2D, accum_abs=None, accum_abs_max=None):
    """The training loop's statements, verbatim in meaning (in place)."""
    visibility_filter = radii > 0
    max_radii2D[visibility_filter] = torch.max(max_radii2D[visibility_filter], radii[visibility_filter])
    xyz_grad[visibility_filter] += torch.norm(grad[visibility_filter, :2], dim=-1, keepdim=True)
    if accum_abs is not None:
        accum_abs[visibility_filter] += torch.norm(grad[visibility_filter, 2:], dim=-1, keepdim=True)
        accum_abs_max[visibility_filter] = torch.max(accum_abs_max[visibility_filter], torch.norm(grad[visibility_filter, 2:], dim=-1, keepdim=True))
    denom[visibility_filter] += 1


def test_restatement_follows_method_py():
    ref = "/root/reference/wildgaussians/method.py"
    if not os.path.isfile(ref):
        pytest.skip("reference checkout not present")
    src = open(ref).read()
    for frag in ("self.xyz_grad[update_filter] += torch.norm(viewspace_point_tensor.grad[update_filter,:2], dim=-1, keepdim=True)",
                 "self.xyz_gradient_accum_abs[update_filter] += torch.norm(viewspace_point_tensor.grad[update_filter,2:], dim=-1, keepdim=True)",
                 "self.xyz_gradient_accum_abs_max[update_filter] = torch.max(self.xyz_gradient_accum_abs_max[update_filter], torch.norm(viewspace_point_tensor.grad[update_filter,2:], dim=-1, keepdim=True))",
                 "self.denom[update_filter] += 1",
                 "self.model.max_radii2D[visibility_filter] = torch.max(self.model.max_radii2D[visibility_filter], radii[visibility_filter])"):
        assert frag in src, frag


def test_densify_abi_exported():
    lib = C.CDLL(os.path.join(ROOT, "wild-gaussians_amd", "diff_gaussian_rasterization", "libwg_rasterizer.so"))
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "wg_densify.h")).read(), flags=re.S)
    assert set(re.findall(r"\b(wg_[a-z0-9_]+)\s*\(", text)) == {"wg_densification_stats"}
    f = lib.wg_densification_stats
    f.restype, f.argtypes = C.c_int, [C.c_int] + [C.c_void_p] * 8
    assert f(-1, None, None, None, None, None, None, None, None) == -1
    assert f(0, None, None, None, None, None, None, None, None) == 0
    assert f(8, None, None, None, None, None, None, None, None) == -1
    from wg_fused_gaussians import add_densification_stats
    with pytest.raises(RuntimeError, match="no CPU path"):
        add_densification_stats(torch.zeros(4, dtype=torch.int32), torch.zeros(4, 3), torch.zeros(4, 1), torch.zeros(4, 1))


@pytest.mark.gpu
@pytest.mark.parametrize("P,gof,with_radii", [(1, True, True), (255, True, True), (4097, False, True), (100000, True, False), (300001, True, True)])
def test_fused_densification_stats_match_the_training_loop(P, gof, with_radii):
    from wg_fused_gaussians import add_densification_stats
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(P)
    radii = (torch.randint(-2, 40, (P,), generator=g).clamp_min(0) * (torch.rand(P, generator=g) < 0.6)).to(torch.int32).to(dev)
    grad = (torch.randn(P, 3, generator=g) * 1e-3).to(dev)
    grad[:, 2] = grad[:, 2].abs()
    mk = lambda: (torch.rand(P, 1, generator=g) * 5e-3).to(dev)
    state = dict(xyz_grad=mk(), denom=torch.randint(0, 9, (P, 1), generator=g).float().to(dev),
                 max_radii2D=torch.randint(0, 30, (P,), generator=g).float().to(dev), accum_abs=mk(), accum_abs_max=mk())
    a = {k: v.clone() for k, v in state.items()}
    b = {k: v.clone() for k, v in state.items()}
    for _ in range(2):  # twice: the buffers accumulate
        ref_stats(radii, grad, a["xyz_grad"], a["denom"], a["max_radii2D"] if with_radii else a["max_radii2D"].clone(),
                  a["accum_abs"] if gof else None, a["accum_abs_max"] if gof else None)
        add_densification_stats(radii, grad, b["xyz_grad"], b["denom"], max_radii2D=b["max_radii2D"] if with_radii else None,
                                xyz_gradient_accum_abs=b["accum_abs"] if gof else None,
                                xyz_gradient_accum_abs_max=b["accum_abs_max"] if gof else None)
    assert torch.equal(a["denom"], b["denom"]) and torch.equal(a["max_radii2D"], b["max_radii2D"])
    assert torch.equal(a["accum_abs_max"], b["accum_abs_max"])
    for k in ("xyz_grad", "accum_abs"):
        assert ((a[k] - b[k]).abs() <= 1e-9 + 1e-6 * a[k].abs()).all(), (k, (a[k] - b[k]).abs().max().item())
    if not gof:
        assert torch.equal(b["accum_abs"], state["accum_abs"])
    if not with_radii:
        assert torch.equal(b["max_radii2D"], state["max_radii2D"])
    untouched = radii <= 0
    assert torch.equal(b["xyz_grad"][untouched], state["xyz_grad"][untouched])


@pytest.mark.gpu
def test_densification_stats_after_a_real_backward_pass():
    """The call pattern of the training loop: the operator's radii and the gradient it leaves in means2D."""
    import numpy as np
    import wg_scenes as S
    from diff_gaussian_rasterization import GaussianRasterizer
    from tests.wg_testlib import make_settings, to_dev
    from wg_fused_gaussians import add_densification_stats
    W, H, P = 320, 180, 20000
    cam, cloud = S.make_camera(W, H), S.make_cloud(P, W, H, sh_degree=1, seed=2, scale_mult=4.0)
    t = {k: to_dev(v) for k, v in cloud.items()}
    viewspace = torch.zeros(P, 3, device="cuda", requires_grad=True)
    color, radii, _ = GaussianRasterizer(make_settings(cam, 1))(means3D=t["means3D"], means2D=viewspace, opacities=t["opacities"], shs=t["shs"],
                                                                scales=t["scales"], rotations=t["rotations"])
    color.backward(to_dev(S.make_cotangent(W, H)))
    z = lambda: torch.zeros(P, 1, device="cuda")
    a = dict(xyz_grad=z(), denom=z(), max_radii2D=torch.zeros(P, device="cuda"), accum_abs=z(), accum_abs_max=z())
    b = {k: v.clone() for k, v in a.items()}
    ref_stats(radii, viewspace.grad, a["xyz_grad"], a["denom"], a["max_radii2D"], a["accum_abs"], a["accum_abs_max"])
    add_densification_stats(radii, viewspace.grad, b["xyz_grad"], b["denom"], b["max_radii2D"], b["accum_abs"], b["accum_abs_max"])
    assert int(b["denom"].sum().item()) == int((radii > 0).sum().item()) > 1000
    for k in a:
        assert ((a[k] - b[k]).abs() <= 1e-12 + 1e-6 * a[k].abs()).all(), k
    assert (b["accum_abs"] > 0).sum() > 1000
