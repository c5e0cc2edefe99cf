#!/usr/bin/env python3
"""Wall-clock duration of every train step of bench.py's workload (host side; every forward waits for the mailbox, so a step's
host time tracks its GPU time): where do the occasional slow timed passes come from?  usage: diag_step_jitter.py [steps] [gc]"""
import gc, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wild-gaussians_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch, wg_scenes as S
from diff_gaussian_rasterization import GaussianRasterizer
from tests.wg_testlib import make_settings, to_dev
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 600
if len(sys.argv) > 2 and sys.argv[2] == "nogc":
    gc.disable()
dev = torch.device("cuda", 0)
P, W, H = 1_000_000, 1920, 1080
cloud = S.make_cloud(P, W, H, sh_degree=3, seed=0); cam = S.make_camera(W, H)
rast = GaussianRasterizer(make_settings(cam, 3, device=dev))
t = {k: to_dev(v, dev).requires_grad_(True) for k, v in cloud.items()}
m2 = torch.zeros((P, 3), device=dev, requires_grad=True)
cot = to_dev(S.make_cotangent(W, H), dev); cf = cot.reshape(-1)
def step():
    for v in t.values(): v.grad = None
    m2.grad = None
    c, r, a = rast(means3D=t["means3D"], means2D=m2, opacities=t["opacities"], shs=t["shs"], scales=t["scales"], rotations=t["rotations"])
    c.backward(cot)
    return torch.dot(c.detach().reshape(-1), cf).reshape(1)
for _ in range(20): step()
torch.cuda.synchronize()
print("loadavg", os.getloadavg(), "cpus", len(os.sched_getaffinity(0)), "gc", gc.isenabled(), gc.get_threshold())
gcs = []
gc.callbacks.append(lambda ph, info: gcs.append((time.perf_counter(), ph, info.get("generation"))))
for rep in range(3):
    ts = np.empty(steps + 1); a0 = torch.cuda.memory_stats()["num_device_alloc"]
    torch.cuda.synchronize(); ts[0] = time.perf_counter()
    for i in range(steps):
        step(); ts[i + 1] = time.perf_counter()
    torch.cuda.synchronize(); tend = time.perf_counter()
    d = np.diff(ts) * 1e3
    med = np.median(d)
    spikes = [(int(i), round(float(x), 2)) for i, x in enumerate(d) if x > 2 * med]
    print(f"pass {rep}: total {1e3*(tend-ts[0])/steps:.4f} ms/step, median {med:.4f}, p90 {np.quantile(d,0.9):.4f}, p99 {np.quantile(d,0.99):.4f}, max {d.max():.2f}, "
          f"sum of excess over median {float((d-med).clip(0).sum()):.1f} ms, device_allocs +{torch.cuda.memory_stats()['num_device_alloc']-a0}")
    print("   spikes (>2x median):", spikes[:30], "gc events in pass:", len([g for g in gcs if g[0] >= ts[0] and g[1] == 'start']), [g[2] for g in gcs if g[0] >= ts[0] and g[1]=='start'][:20])
