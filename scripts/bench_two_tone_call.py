#!/usr/bin/env python3
"""The operator alone on WildGaussians' two renders of one SH block: two toned calls (sh_mul= ...; sh_pre_clamp_max= alone for the raw one)
against ONE two-tone call (sh_second=True), forward + backward, synthetic scene.   usage: bench_two_tone_call.py [P W H steps]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wild-gaussians_amd")); sys.path.insert(0, ROOT)
import torch
import wg_scenes as S
from diff_gaussian_rasterization import GaussianRasterizer
from tests.wg_testlib import make_settings, to_dev

P, W, H, steps = (int(a) for a in (sys.argv[1:5] + ["1000000", "1920", "1080", "200"][len(sys.argv) - 1:]))
dev = torch.device("cuda", 0)
cloud = S.make_cloud(P, W, H, sh_degree=3, seed=0)
rast = GaussianRasterizer(make_settings(S.make_camera(W, H), 3, device=dev))
t = {k: to_dev(v, dev).requires_grad_(True) for k, v in cloud.items()}
mul = (1.0 + 0.01 * torch.randn(P, 3, device=dev)).requires_grad_(True)
off = (0.01 * torch.randn(P, 3, device=dev)).requires_grad_(True)
m2d = torch.zeros((P, 3), device=dev, requires_grad=True)
cot1, cot2 = to_dev(S.make_cotangent(W, H, seed=1), dev), to_dev(S.make_cotangent(W, H, seed=2), dev)
kw = dict(means3D=t["means3D"], means2D=m2d, opacities=t["opacities"], shs=t["shs"], scales=t["scales"], rotations=t["rotations"])
tone = dict(sh_mul=mul, sh_offset=off, sh_pre_clamp_max=1.0, sh_post_clamp_max=1.0)


def clear():
    for v in list(t.values()) + [mul, off, m2d]:
        v.grad = None


def two_calls():
    clear()
    a = rast(**kw, **tone)[0]
    b = rast(**kw, sh_pre_clamp_max=1.0)[0]
    torch.autograd.backward([a, b], [cot1, cot2])


def one_call():
    clear()
    a, _, _, b = rast(**kw, **tone, sh_second=True, sh_pre_clamp_max2=1.0)
    torch.autograd.backward([a, b], [cot1, cot2])


out = {"workload": f"{P} Gaussians, {W}x{H}, SH 3: toned + clamped renders of one SH block, forward + backward"}
for name, fn in (("two_toned_calls_ms", two_calls), ("one_two_tone_call_ms", one_call), ("two_toned_calls_again_ms", two_calls)):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    out[name] = round(1e3 * (time.perf_counter() - t0) / steps, 4)
print(json.dumps(out))
