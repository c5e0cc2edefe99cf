#!/usr/bin/env python3
"""Fused SSIM vs the reference formula in PyTorch: fwd+bwd time at 1600x1200 and the HBM roofline fraction."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wild-gaussians_amd")); sys.path.insert(0, ROOT)
import torch
from wg_fused_ssim import ssim
from tests.test_ssim import ref_ssim
dev = torch.device("cuda", 0)
H, W = 1200, 1600
g = torch.Generator().manual_seed(0)
x = torch.rand(3, H, W, generator=g).to(dev).requires_grad_(True)
y = torch.rand(3, H, W, generator=g).to(dev)
wgt = torch.rand(H, W, generator=g).to(dev)

def run(fn, batched):
    def step():
        x.grad = None
        m = fn(x[None], y[None], size_average=False)[0] if batched else fn(x, y, size_average=False)
        ((1.0 - m) * wgt).mean().backward()
    for _ in range(5): step()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(50): step()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / 50

t_ref, t_fused = run(ref_ssim, True), run(ssim, False)
N = 3 * H * W
bytes_ = N * (8 + 16) + N * (24 + 4)   # forward: 2 images in, map + 3 derivative maps out; backward: 6 maps in, 1 out
print(json.dumps({"workload": f"SSIM map fwd+bwd, 3x{H}x{W} float32", "reference_formula_torch_ms": round(t_ref, 4),
                  "fused_ms (incl. the weighting / mean torch ops)": round(t_fused, 4), "speedup": round(t_ref / t_fused, 1),
                  "algorithmic_bytes": bytes_, "note": "kernel-only times: rocprofv3 summary"}))
