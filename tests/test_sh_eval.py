"""Fused eval_sh (include/wg_sh_eval.h, wg_fused_gaussians.eval_sh) against the reference's formula
(wildgaussians/method.py:493-548): the real function when the staged caller is present, and a float64 torch statement of the same
real-SH polynomials -- values and, through autograd, both gradients."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

C0, C1 = 0.28209479177387814, 0.4886025119029199
C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658, 1.445305721320277,
      -0.5900435899266435]


def basis(deg, d):
    """[P, (deg+1)^2] real SH basis in the reference's convention."""
    x, y, z = d[:, 0], d[:, 1], d[:, 2]
    b = [torch.full_like(x, C0)]
    if deg > 0:
        b += [-C1 * y, C1 * z, -C1 * x]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        b += [C2[0] * xy, C2[1] * yz, C2[2] * (2 * zz - xx - yy), C2[3] * xz, C2[4] * (xx - yy)]
    if deg > 2:
        b += [C3[0] * y * (3 * xx - yy), C3[1] * xy * z, C3[2] * y * (4 * zz - xx - yy), C3[3] * z * (2 * zz - 3 * xx - 3 * yy),
              C3[4] * x * (4 * zz - xx - yy), C3[5] * z * (xx - yy), C3[6] * x * (xx - 3 * yy)]
    return torch.stack(b, dim=1)


def restated(deg, sh, dirs):
    n = (deg + 1) ** 2
    return torch.einsum("pck,pk->pc", sh[..., :n], basis(deg, dirs))


@pytest.mark.parametrize("K", [16, 9, 25])
@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_fused_eval_sh_matches_the_formula_and_its_gradients(deg, K):
    if K < (deg + 1) ** 2:
        pytest.skip("fewer coefficients than the degree needs")
    from wg_fused_gaussians import eval_sh
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(deg * 100 + K)
    P = 4099
    sh = torch.randn(P, 3, K, generator=g).to(dev).requires_grad_(True)
    dirs = torch.nn.functional.normalize(torch.randn(P, 3, generator=g), dim=1).to(dev).requires_grad_(True)
    cot = torch.randn(P, 3, generator=g).to(dev)
    out = eval_sh(deg, sh, dirs)
    out.backward(cot)
    sh64, d64 = sh.detach().double().requires_grad_(True), dirs.detach().double().requires_grad_(True)
    ref = restated(deg, sh64, d64)
    ref.backward(cot.double())
    tol = lambda t: 3e-6 * (float(t.abs().max()) + 1e-30)
    assert float((out.detach().double() - ref.detach()).abs().max()) <= tol(ref.detach())
    assert float((sh.grad.double() - sh64.grad).abs().max()) <= tol(sh64.grad)
    ref_dgrad = d64.grad if d64.grad is not None else torch.zeros_like(d64)   # degree 0 does not depend on the direction
    assert float((dirs.grad.double() - ref_dgrad).abs().max()) <= tol(ref_dgrad)
    assert not sh.grad[..., (deg + 1) ** 2:].any()   # coefficients beyond the active degree take exactly zero gradient


def test_fused_eval_sh_against_the_reference_function_itself_and_its_call_shapes():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "real_caller"))
    import harness
    if not harness.staged_available():
        pytest.skip("no reference checkout and nothing staged")
    m = harness.import_method()
    from wg_fused_gaussians import eval_sh
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(9)
    P = 20000
    feats = torch.randn(P, 48, generator=g).to(dev).requires_grad_(True)
    xyz = torch.randn(P, 3, generator=g).to(dev).requires_grad_(True)
    cam = torch.tensor([0.3, -0.2, 4.0], device=dev)
    cot = torch.randn(P, 3, generator=g).to(dev)

    def colours(fn, deg):   # the caller's statements around the function (method.py:1558-1565)
        d = torch.nn.functional.normalize(xyz - cam.repeat(P, 1), dim=1)
        view = feats.clamp_max(1.0).view(-1, 16, 3).transpose(1, 2).contiguous()
        return torch.clamp_min(fn(deg, view, d) + 0.5, 0.0)
    for deg in (torch.tensor(3, device=dev), 1):   # the caller passes its `active_sh_degree` buffer: a tensor
        res = []
        for fn in (m.eval_sh, eval_sh):
            feats.grad = xyz.grad = None
            c = colours(fn, deg)
            c.backward(cot)
            res.append((c.detach().clone(), feats.grad.clone(), xyz.grad.clone()))
        for a, b in zip(*res):
            assert float((a - b).abs().max()) <= 2e-5 * (float(a.abs().max()) + 1e-30)
    with pytest.raises(RuntimeError, match="no CPU path"):
        eval_sh(1, torch.zeros(4, 3, 4), torch.zeros(4, 3))
    with pytest.raises(NotImplementedError):
        eval_sh(4, torch.zeros(4, 3, 25, device=dev), torch.zeros(4, 3, device=dev))
