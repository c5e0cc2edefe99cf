"""wg_fused_gaussians.tall_linear: same values and gradients as torch.nn.functional.linear (it is plain PyTorch: the weight
gradient of a millions-of-rows input as one batched product over row chunks -- INTEGRATION.md 5)."""
import os
import sys

import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wild-gaussians_amd"))


@pytest.mark.parametrize("n,cin,cout,chunks,bias", [(1000, 59, 128, 16, True), (1003, 128, 128, 16, True), (37, 128, 6, 256, False), (5, 3, 2, 1, True)])
def test_tall_linear_equals_linear(n, cin, cout, chunks, bias):
    from wg_fused_gaussians import tall_linear
    g = torch.Generator().manual_seed(n)
    x0, w0 = torch.randn(n, cin, generator=g, dtype=torch.float64), torch.randn(cout, cin, generator=g, dtype=torch.float64)
    b0 = torch.randn(cout, generator=g, dtype=torch.float64) if bias else None
    gy = torch.randn(n, cout, generator=g, dtype=torch.float64)
    outs = []
    for fn in (lambda x, w, b: F.linear(x, w, b), lambda x, w, b: tall_linear(x, w, b, chunks=chunks)):
        x, w = x0.clone().requires_grad_(True), w0.clone().requires_grad_(True)
        b = None if b0 is None else b0.clone().requires_grad_(True)
        y = fn(x, w, b)
        y.backward(gy)
        outs.append((y.detach(), x.grad, w.grad, None if b is None else b.grad))
    for a, c in zip(*outs):
        if a is not None:
            assert torch.allclose(a, c, rtol=1e-12, atol=1e-12)


def test_tall_linear_only_weight_needs_grad():
    from wg_fused_gaussians import tall_linear
    x, w = torch.randn(64, 8), torch.randn(4, 8, requires_grad=True)
    tall_linear(x, w, None, chunks=8).sum().backward()
    assert torch.allclose(w.grad, x.sum(0)[None].expand(4, 8), atol=1e-5)
