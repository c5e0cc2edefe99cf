#!/usr/bin/env python3
"""The kernels either side of the path against the HBM roofline, at config 3's sizes (3 M Gaussians, 1600x1200):
fused Adam over WildGaussians' parameter groups, fused activations fwd / bwd, fused eval_sh fwd / bwd, densification statistics, fused
L1 + DSSIM loss fwd / bwd.  Streaming kernels: algorithmic bytes = every input read once + every output written once.

usage: python scripts/bench_optins.py [gaussians width height iters]     -> one JSON line (ms, GB/s, fraction of 8 TB/s per kernel)
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wild-gaussians_amd"))
import torch  # noqa: E402
import wg_fused_gaussians as FG  # noqa: E402
import wg_fused_ssim as FS  # noqa: E402

P, W, H, N = (int(a) for a in (sys.argv[1:5] + ["3000000", "1600", "1200", "30"][len(sys.argv) - 1:]))
dev = torch.device("cuda", 0)
PEAK = 8000.0
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: torch.randn(*s, device=dev, generator=g)  # noqa: E731


def timed(fn, iters=N, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)
    return ms[len(ms) // 2]


def row(name, ms, nbytes):
    gbps = nbytes / ms / 1e6
    return {"kernel": name, "ms": round(ms, 4), "algorithmic_MB": round(nbytes / 1e6, 1), "GBps": round(gbps, 1), "frac_of_8TBps": round(gbps / PEAK, 3)}


rows = []

# ---- Adam: WildGaussians' groups (method.py:1030-1049): xyz 3, features_dc 3, features_rest 45, opacity 1, scales 3, rotations 4,
#      appearance embeddings 24 floats per Gaussian; 16 B read + 12 B written per element
shapes = [(P, 3), (P, 1, 3), (P, 15, 3), (P, 1), (P, 3), (P, 4), (P, 24)]
params = [torch.nn.Parameter(rnd(*s) * 0.1) for s in shapes]
opt = FG.FusedAdam([{"params": [p], "lr": 1e-3} for p in params], eps=1e-15)
for p in params:
    p.grad = rnd(*p.shape) * 1e-3
opt.step()
numel = sum(p.numel() for p in params)
rows.append(row("fused_adam (7 groups, %d floats per Gaussian)" % (numel // P), timed(opt.step), numel * 28))

# ---- activations forward / backward: reads 9 floats, writes 8 (fwd); reads 9 + 8 cotangents, writes 8 (bwd)
ro, rs, rr, f3 = rnd(P, 1).requires_grad_(True), (rnd(P, 3) * 0.3 - 4.0).requires_grad_(True), rnd(P, 4).requires_grad_(True), rnd(P, 1).abs() * 0.01
rows.append(row("activations_forward", timed(lambda: FG.activate(ro.detach(), rs.detach(), rr.detach(), f3)), P * (9 + 8) * 4))
op, sc, rot = FG.activate(ro, rs, rr, f3)
gop, gsc, grot = rnd(P, 1), rnd(P, 3), rnd(P, 4)


def act_bwd():
    torch.autograd.grad((op, sc, rot), (ro, rs, rr), (gop, gsc, grot), retain_graph=True)


rows.append(row("activations_backward", timed(act_bwd), P * (9 + 8 + 8) * 4))

# ---- eval_sh (degree 3): reads 48 + 3 floats, writes 3 (fwd); reads 48 + 3 + 3, writes 48 + 3 (bwd)
sh = (rnd(P, 3, 16) * 0.1).requires_grad_(True)
dirs = torch.nn.functional.normalize(rnd(P, 3), dim=1).requires_grad_(True)
rows.append(row("eval_sh_forward<3>", timed(lambda: FG.eval_sh(3, sh.detach(), dirs.detach())), P * (48 + 3 + 3) * 4))
rgb = FG.eval_sh(3, sh, dirs)
grgb = rnd(P, 3)
rows.append(row("eval_sh_backward<3>", timed(lambda: torch.autograd.grad(rgb, (sh, dirs), grgb, retain_graph=True)), P * (48 + 3 + 3 + 48 + 3) * 4))

# ---- densification statistics: radii 4 + viewspace grad 12 read, xyz_grad 4 + denom 4 + max_radii 4 + abs 4 + abs_max 4 read and written
radii = torch.randint(0, 40, (P,), device=dev, dtype=torch.int32)
vg = rnd(P, 3)
acc = [torch.zeros(P, 1, device=dev) for _ in range(2)] + [torch.zeros(P, device=dev)] + [torch.zeros(P, 1, device=dev) for _ in range(2)]
rows.append(row("densification_stats", timed(lambda: FG.add_densification_stats(radii, vg, acc[0], acc[1], max_radii2D=acc[2], xyz_gradient_accum_abs=acc[3],
                                                                                   xyz_gradient_accum_abs_max=acc[4])), P * (16 + 5 * 8)))

# ---- fused L1 + DSSIM loss: three [3,H,W] images read, (bwd) two gradients written
img = torch.rand(3, H, W, device=dev, generator=g).requires_grad_(True)
gt = torch.rand(3, H, W, device=dev, generator=g)
rows.append(row("l1_ssim_loss_forward", timed(lambda: FS.l1_ssim_loss(img.detach(), img.detach(), gt)), 3 * H * W * 4 * 2))
loss = FS.l1_ssim_loss(img, img, gt)
rows.append(row("l1_ssim_loss_backward", timed(lambda: torch.autograd.grad(loss, img, retain_graph=True)), 3 * H * W * 4 * 3))

print(json.dumps({"workload": f"{P} Gaussians, {W}x{H}; median of {N} calls, HIP events around each call (binding overhead included)",
                  "library": os.environ.get("WG_RASTERIZER_LIB", "in-tree"), "rows": rows}))
