#!/usr/bin/env python3
"""Where does the non-temporal SH stream's gain at 1 M Gaussians come from?  Steps (fwd + bwd, one event pair each) with the option on / off,
(a) back to back as bench.py runs them -- the SAME parameter tensors every step -- and (b) with 1.2 GB of unrelated traffic between two steps
(what an optimizer pass over the parameters does to the caches in a training loop).  Usage: python scripts/r6/diag_sh_stream.py [gaussians]"""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "wild-gaussians_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
import wg_scenes as S
from diff_gaussian_rasterization import GaussianRasterizer, _C
from tests.wg_testlib import make_settings, to_dev

P = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
W, H = 1920, 1080
dev = torch.device("cuda", 0)
_C.set_option("geometry_reuse", 0)
cloud, cam, cot = S.make_cloud(P, W, H, sh_degree=3, seed=0), S.make_camera(W, H), to_dev(S.make_cotangent(W, H), dev)
rast = GaussianRasterizer(make_settings(cam, 3, device=dev))
t = {k: to_dev(v, dev).requires_grad_(True) for k, v in cloud.items()}
means2D = torch.zeros((P, 3), device=dev, requires_grad=True)
junk = torch.zeros(150_000_000, device=dev)   # 600 MB: read + written between steps in mode (b)

def step():
    for v in t.values():
        v.grad = None
    means2D.grad = None
    color, radii, acc = rast(means3D=t["means3D"], means2D=means2D, opacities=t["opacities"], shs=t["shs"], colors_precomp=None, scales=t["scales"], rotations=t["rotations"])
    color.backward(cot)

def run(flush, n=200):
    import gc
    gc.collect(); gc.disable()
    for _ in range(30):
        step()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    _C.profile_reset(); _C.profile_enable(True)
    torch.cuda.synchronize(dev)
    for a, b in evs:
        if flush:
            junk.add_(1.0)
        a.record(); step(); b.record()
    torch.cuda.synchronize(dev)
    st = _C.profile_read(); _C.profile_enable(False)
    gc.enable()
    ms = sorted(a.elapsed_time(b) for a, b in evs)
    avg = lambda k: round(st[k][0] / max(1, st[k][1]), 4) if k in st else None
    return {"step_ms_p50": round(ms[n // 2], 4), **{k: avg(k) for k in ("preprocess", "sort", "render_forward", "render_backward", "preprocess_backward")}}

out = {"workload": f"{P} Gaussians, {W}x{H}, SH 3, fwd+bwd"}
for flush in (False, True):
    for mode in (0, 1, 0, 1):
        _C.set_option("sh_stream", mode)
        out.setdefault("steps back to back" if not flush else "1.2 GB of other traffic between steps", []).append({"sh_stream": mode, **run(flush)})
print(json.dumps(out, indent=1))
