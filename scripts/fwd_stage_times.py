#!/usr/bin/env python3
"""Forward-only stage times of the bench scene under the current WG_OPTIONS (tuning aid)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wild-gaussians_amd")); sys.path.insert(0, ROOT)
import torch, wg_scenes as S
from diff_gaussian_rasterization import GaussianRasterizer, _C
from tests.wg_testlib import make_settings, to_dev
dev = torch.device("cuda", 0)
P, W, H = 1000000, 1920, 1080
cloud = S.make_cloud(P, W, H, sh_degree=3, seed=0); cam = S.make_camera(W, H)
rs = make_settings(cam, 3, device=dev); rast = GaussianRasterizer(rs)
t = {k: to_dev(v, dev) for k, v in cloud.items()}
m2 = torch.zeros_like(t["means3D"])
def f():
    with torch.no_grad():
        return rast(means3D=t["means3D"], means2D=m2, opacities=t["opacities"], shs=t["shs"], scales=t["scales"], rotations=t["rotations"])
for _ in range(10): f()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(100): f()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 100
_C.profile_reset(); _C.profile_enable(True)
for _ in range(30): f()
torch.cuda.synchronize()
st = _C.profile_read(); _C.profile_enable(False)
print(os.environ.get("WG_OPTIONS", ""), "fwd_ms %.4f" % (dt * 1e3), json.dumps({k: round(ms / n, 4) for k, (ms, n) in st.items() if n}))
