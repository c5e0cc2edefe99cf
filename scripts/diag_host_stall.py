#!/usr/bin/env python3
"""Host-side duration of each forward call vs the GPU time (diagnostic for allocation / driver stalls)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wild-gaussians_amd")); sys.path.insert(0, ROOT)
import torch, wg_scenes as S
from diff_gaussian_rasterization import GaussianRasterizer
from tests.wg_testlib import make_settings, to_dev
dev = torch.device("cuda", 0)
P, W, H = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
cloud = S.make_cloud(P, W, H, sh_degree=3, seed=0); cam = S.make_camera(W, H)
rs = make_settings(cam, 3, device=dev); rast = GaussianRasterizer(rs)
t = {k: to_dev(v, dev) for k, v in cloud.items()}
m2 = torch.zeros_like(t["means3D"])
def f():
    with torch.no_grad():
        return rast(means3D=t["means3D"], means2D=m2, opacities=t["opacities"], shs=t["shs"], scales=t["scales"], rotations=t["rotations"])
for _ in range(3): f()
torch.cuda.synchronize()
for i in range(12):
    st = torch.cuda.memory_stats()
    a0, f0 = st["num_device_alloc"], st["num_device_free"]
    t0 = time.perf_counter(); f(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    st = torch.cuda.memory_stats()
    print(f"iter {i}: call {1e3*(t1-t0):.2f} ms, sync {1e3*(t2-t1):.2f} ms, device_alloc +{st['num_device_alloc']-a0} free +{st['num_device_free']-f0}, reserved {st['reserved_bytes.all.current']/2**30:.2f} GiB")
