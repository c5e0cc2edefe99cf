#!/usr/bin/env python3
"""How long does the GPU need to regain its clocks after sitting idle?  Headline fwd+bwd steps, per-step HIP-event times, after a host
sleep of 0 / 2 / 10 / 50 / 200 ms (GPU idle), and the time a gc.collect() takes in this process.  One JSON line.
(round 4: bench.py's 20-step timed region read 7 % above its p50; VERDICT r3 "weak" item 4)"""
import gc, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wild-gaussians_amd")); sys.path.insert(0, ROOT)
import torch
import wg_scenes as S
from diff_gaussian_rasterization import GaussianRasterizer
from tests.wg_testlib import make_settings, to_dev

W, H, P = 1920, 1080, 1_000_000
dev = torch.device("cuda", 0)
cloud = S.make_cloud(P, W, H, sh_degree=3, seed=0)
rast = GaussianRasterizer(make_settings(S.make_camera(W, H), 3, device=dev))
t = {k: to_dev(v, dev).requires_grad_(True) for k, v in cloud.items()}
m2d = torch.zeros((P, 3), device=dev, requires_grad=True)
cot = to_dev(S.make_cotangent(W, H), dev)


def step():
    for v in t.values():
        v.grad = None
    m2d.grad = None
    rast(means3D=t["means3D"], means2D=m2d, opacities=t["opacities"], shs=t["shs"], scales=t["scales"], rotations=t["rotations"])[0].backward(cot)


for _ in range(100):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter(); gc.collect(); gc_ms = 1e3 * (time.perf_counter() - t0)
out = {"gc_collect_ms": round(gc_ms, 2), "after_idle_ms": {}}
for idle in (0, 2, 10, 50, 200, 0):
    for _ in range(60):
        step()
    torch.cuda.synchronize()
    time.sleep(idle / 1e3)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(40)]
    for a, b in evs:
        a.record(); step(); b.record()
    torch.cuda.synchronize()
    ms = [round(a.elapsed_time(b), 3) for a, b in evs]
    out["after_idle_ms"].setdefault(str(idle), []).append({"first_10": ms[:10], "steps_10_19_mean": round(sum(ms[10:20]) / 10, 3), "steps_30_39_mean": round(sum(ms[30:]) / 10, 3)})
print(json.dumps(out))
