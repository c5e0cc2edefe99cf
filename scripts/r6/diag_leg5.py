#!/usr/bin/env python3
"""Why does bench.py's config-5 leg read 2.00 ms per frame where the stand-alone run reads 1.84?  The leg's loop in a process of its own, with and
without a headline workload in front of it; per-100-frame times and the near aim.  Usage: python scripts/r6/diag_leg5.py [--headline-first]"""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "wild-gaussians_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
import wg_scenes as S
from diff_gaussian_rasterization import GaussianRasterizer, _C
from tests.wg_testlib import make_settings, to_dev
dev = torch.device("cuda", 0)
_C.set_option("geometry_reuse", 0)
out = {"argv": sys.argv[1:]}
if "--headline-first" in sys.argv:
    W, H, P = 1920, 1080, 1_000_000
    cloud, cam, cot = S.make_cloud(P, W, H, sh_degree=3, seed=0), S.make_camera(W, H), to_dev(S.make_cotangent(W, H), dev)
    rast = GaussianRasterizer(make_settings(cam, 3, device=dev))
    t = {k: to_dev(v, dev).requires_grad_(True) for k, v in cloud.items()}
    m2 = torch.zeros((P, 3), device=dev, requires_grad=True)
    for _ in range(100):
        for v in t.values():
            v.grad = None
        m2.grad = None
        c = rast(means3D=t["means3D"], means2D=m2, opacities=t["opacities"], shs=t["shs"], colors_precomp=None, scales=t["scales"], rotations=t["rotations"])[0]
        c.backward(cot)
    torch.cuda.synchronize(dev)
    del t, m2, rast, cloud, c
    import gc; gc.collect(); torch.cuda.empty_cache()
W, H, P = 3840, 2160, 10_000_000
cloud, cam = S.make_cloud(P, W, H, sh_degree=3, seed=0), S.make_camera(W, H)
rast = GaussianRasterizer(make_settings(cam, 3, device=dev))
t = {k: to_dev(v, dev) for k, v in cloud.items()}
m2 = torch.zeros((P, 3), device=dev)
def step():
    with torch.no_grad():
        return rast(means3D=t["means3D"], means2D=m2, opacities=t["opacities"], shs=t["shs"], colors_precomp=None, scales=t["scales"], rotations=t["rotations"])[0]
if "--gc-off" in sys.argv:
    import gc; gc.collect(); gc.disable()
for _ in range(64):
    step()
torch.cuda.synchronize(dev)
rows = []
for blk in range(8):
    t0 = time.perf_counter()
    for _ in range(100):
        step()
    torch.cuda.synchronize(dev)
    rows.append({"ms_per_frame": round((time.perf_counter() - t0) * 10, 4), "aim": _C.get_option("near_per_tile_now"), "floor": _C.get_option("near_floor_now"),
                 "far_last": _C.get_option("near_far_tiles_last"), "backoff": _C.get_option("near_split_backoff"), "spec_misses": _C.get_option("spec_misses")})
out["blocks_of_100_frames"] = rows
print(json.dumps(out))
