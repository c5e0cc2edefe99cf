#!/usr/bin/env python3
"""Do the sporadic multi-millisecond steps of the fwd+bwd loop need autograd's worker thread?  Alternating blocks of the SAME step made two
ways on the same box: (A) as a training loop makes it -- GaussianRasterizer(...) + image.backward(cotangent): the backward launches come
from autograd's device thread, the main thread waits for it -- and (B) the binding's two entry points called directly from the main thread
(_C.rasterize_gaussians / _C.rasterize_gaussians_backward: the reference's own tests call its extension this way), no second thread.
Per mode: steps, median, mean, and every step above 1.5 x the median.   usage: diag_stall_threads.py [blocks=10] [steps_per_block=2000]"""
import gc, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wild-gaussians_amd")); sys.path.insert(0, ROOT)
import torch
import wg_scenes as S
from diff_gaussian_rasterization import GaussianRasterizer, _C
from tests.wg_testlib import make_settings, to_dev

blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 10
per = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
W, H, P = 1920, 1080, 1_000_000
dev = torch.device("cuda", 0)
cloud = S.make_cloud(P, W, H, sh_degree=3, seed=0)
rs = make_settings(S.make_camera(W, H), 3, device=dev)
rast = GaussianRasterizer(rs)
t = {k: to_dev(v, dev).requires_grad_(True) for k, v in cloud.items()}
m2d = torch.zeros((P, 3), device=dev, requires_grad=True)
cot = to_dev(S.make_cotangent(W, H), dev)
E = torch.Tensor([])
d = {k: v.detach() for k, v in t.items()}


def step_autograd():
    for v in t.values():
        v.grad = None
    m2d.grad = None
    rast(means3D=t["means3D"], means2D=m2d, opacities=t["opacities"], shs=t["shs"], scales=t["scales"], rotations=t["rotations"])[0].backward(cot)


def step_direct():
    R, color, radii, gb, bb, ib = _C.rasterize_gaussians(rs.bg, d["means3D"], E, d["opacities"], d["scales"], d["rotations"], rs.scale_modifier, E, rs.viewmatrix,
                                                         rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.kernel_size, rs.subpixel_offset, rs.image_height,
                                                         rs.image_width, d["shs"], rs.sh_degree, rs.campos, rs.prefiltered, False)
    return _C.rasterize_gaussians_backward(rs.bg, d["means3D"], radii, E, d["scales"], d["rotations"], rs.scale_modifier, E, rs.viewmatrix, rs.projmatrix,
                                           rs.tanfovx, rs.tanfovy, rs.kernel_size, rs.subpixel_offset, cot, d["shs"], rs.sh_degree, rs.campos, gb, R, bb, ib, False)


gc.collect(); gc.disable()
for fn in (step_autograd, step_direct):
    for _ in range(100):
        fn()
torch.cuda.synchronize()
rec = {"autograd": [], "direct": []}
for b in range(blocks):
    for name, fn in (("autograd", step_autograd), ("direct", step_direct)):
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        prev = time.perf_counter()
        for _ in range(per):
            fn()
            now = time.perf_counter()
            rec[name].append(now - prev)
            prev = now
        torch.cuda.synchronize()
out = {"box_loadavg": open("/proc/loadavg").read().strip()}
for name, v in rec.items():
    ms = [1e3 * x for x in v]
    med = sorted(ms)[len(ms) // 2]
    slow = [(i, round(x, 3)) for i, x in enumerate(ms) if x > 1.5 * med]
    out[name] = {"steps": len(ms), "median_ms": round(med, 4), "mean_ms": round(sum(ms) / len(ms), 4), "n_over_1.5x_median": len(slow),
                 "excess_ms_total": round(sum(x - med for _, x in slow), 2), "slow": slow[:40]}
print(json.dumps(out))
