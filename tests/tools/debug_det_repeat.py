#!/usr/bin/env python3
"""Debugging aid: ONE forward call, then the deterministic backward several times on that frame's buffers (what retain_graph=True does);
reports which gradient tensors differ between consecutive runs and which byte ranges of the three scratch buffers changed in between.
usage: python tests/tools/debug_det_repeat.py [P W H deg runs scale_mult]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "wild-gaussians_amd")); sys.path.insert(0, ROOT)
import numpy as np
import torch
import wg_scenes as S
from diff_gaussian_rasterization import _C
from tests.wg_testlib import make_settings, to_dev

P, W, H, deg, runs = (int(a) for a in (sys.argv[1:6] + ["20000", "320", "200", "1", "5"][len(sys.argv[1:6]):]))
SCALE = float(sys.argv[6]) if len(sys.argv) > 6 else 2.0
dev = "cuda"
cloud = S.make_cloud(P, W, H, sh_degree=deg, seed=5, scale_mult=SCALE)
cam = S.make_camera(W, H)
cot = to_dev(S.make_cotangent(W, H), dev)
rs = make_settings(cam, deg, device=dev)
e = torch.Tensor([])
t = {k: to_dev(v, dev) for k, v in cloud.items()}
side = torch.cuda.Stream() if os.environ.get("WG_DEBUG_SIDE_STREAM") else None
if side is not None:
    torch.cuda.synchronize()
    torch.cuda.set_stream(side)
    print("on a side stream")
for _ in range(int(os.environ.get("WG_DEBUG_FORWARDS", "1"))):
    R, color, radii, gb, bb, ib = _C.rasterize_gaussians(rs.bg, t["means3D"], e, t["opacities"], t["scales"], t["rotations"], 1.0, e, rs.viewmatrix, rs.projmatrix,
                                                         rs.tanfovx, rs.tanfovy, rs.kernel_size, rs.subpixel_offset, rs.image_height, rs.image_width, t["shs"], deg,
                                                         rs.campos, False, False)
print("R", R, "buffers", gb.numel(), bb.numel(), ib.numel())


def backward(det):
    return _C.rasterize_gaussians_backward(rs.bg, t["means3D"], radii, e, t["scales"], t["rotations"], 1.0, e, rs.viewmatrix, rs.projmatrix, rs.tanfovx,
                                           rs.tanfovy, rs.kernel_size, rs.subpixel_offset, cot, t["shs"], deg, rs.campos, gb, R, bb, ib, False,
                                           options=dict(deterministic_backward=det))


names = ["dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales", "dL_drotations"]
plain = [g.clone() for g in backward(0)]
torch.cuda.synchronize()
snap = [b.clone() for b in (gb, bb, ib)]
prev = None
for r in range(runs):
    g = [x.clone() for x in backward(1)]
    torch.cuda.synchronize()
    now = [b.clone() for b in (gb, bb, ib)]
    for nm, a, b in zip(("geometry", "binning", "image"), snap, now):
        d = (a != b).nonzero().flatten()
        if d.numel():
            print(f"  run {r}: {nm} buffer changed in {d.numel()} bytes, range [{int(d.min())}, {int(d.max())}]")
    snap = now
    worst = max(float((a - b).abs().max() / (b.abs().max() + 1e-30)) for a, b in zip(g, plain) if a.numel())
    msg = f"run {r}: vs the atomic sums {worst:.2e}"
    if prev is not None:
        diff = {nm: int((a != b).sum()) for nm, a, b in zip(names, g, prev) if a.numel() and (a != b).any()}
        msg += f"; vs run {r - 1}: {diff if diff else 'bit-identical'}"
    print(msg)
    prev = g
v = _C.view_geometry(gb, P)
print("view_geometry keys", list(v.keys()))
