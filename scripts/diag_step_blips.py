#!/usr/bin/env python3
"""Which steps of a long fwd+bwd loop are slow, and how regularly?  Headline scene, N steps, host clock after every step (the host is in
step with the GPU); prints the indices and durations of the steps above 1.15 x the median and the gaps between them.
Run it twice: as it is, and with HSA_KERNARG_POOL_SIZE=<bytes> in the environment (ROCclr's kernel-argument ring: when it wraps the
runtime waits for every launch in flight -- a queue drain once per ring).  usage: python scripts/diag_step_blips.py [steps]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wild-gaussians_amd")); sys.path.insert(0, ROOT)
import torch
import wg_scenes as S
from diff_gaussian_rasterization import GaussianRasterizer
from tests.wg_testlib import make_settings, to_dev
import gc

N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
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


gc.collect(); gc.disable()
for _ in range(100):
    step()
torch.cuda.synchronize()
st = [time.perf_counter()]
for _ in range(N):
    step()
    st.append(time.perf_counter())
torch.cuda.synchronize()
d = [1e3 * (b - a) for a, b in zip(st[:-1], st[1:])]
med = sorted(d)[len(d) // 2]
slow = [(i, round(x, 3)) for i, x in enumerate(d) if x > 1.15 * med]
idx = [i for i, _ in slow]
print(json.dumps({"HSA_KERNARG_POOL_SIZE": os.environ.get("HSA_KERNARG_POOL_SIZE"), "steps": N, "median_ms": round(med, 4), "mean_ms": round(sum(d) / len(d), 4),
                  "slow_steps": slow[:60], "gaps_between_slow_steps": [b - a for a, b in zip(idx[:-1], idx[1:])][:60]}))
