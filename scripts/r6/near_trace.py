#!/usr/bin/env python3
"""Per-frame trace of the near / far split's adaptive aim (api.hip: NearAdapt) at BASELINE config 5's size: frame time (host clock around a
synchronized forward), the aim, its floor, the tiles that asked for far instances in the last reported frame, the back-off counter.
Printed run-length encoded.  Usage: python scripts/r6/near_trace.py [frames] [gaussians] [W] [H]"""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "wild-gaussians_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
import wg_scenes as S
from diff_gaussian_rasterization import GaussianRasterizer, _C
from tests.wg_testlib import make_settings, to_dev

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
P = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
W = int(sys.argv[3]) if len(sys.argv) > 3 else 3840
H = int(sys.argv[4]) if len(sys.argv) > 4 else 2160
dev = torch.device("cuda", 0)
_C.set_option("geometry_reuse", 0)
cloud = S.make_cloud(P, W, H, sh_degree=3, seed=0)
cam = S.make_camera(W, H)
rast = GaussianRasterizer(make_settings(cam, 3, device=dev))
t = {k: to_dev(v, dev) for k, v in cloud.items()}
means2D = torch.zeros((P, 3), device=dev)
rows = []
# bench.py's order of passes: K frames with the library's per-stage events on, then warm-up + K plain frames; no synchronize inside a pass (the
# host clock after every call: a forward call returns behind its frame's scan, the host is in step with the GPU)
bench_like = os.environ.get("NEAR_TRACE_BENCH_LIKE", "0") != "0"
def frame():
    rast(means3D=t["means3D"], means2D=means2D, opacities=t["opacities"], shs=t["shs"], colors_precomp=None, scales=t["scales"], rotations=t["rotations"])
def sample(ms):
    rows.append((round(ms, 1), _C.get_option("near_per_tile_now"), _C.get_option("near_floor_now"), _C.get_option("near_far_tiles_last"), _C.get_option("near_split_backoff")))
with torch.no_grad():
    if bench_like:
        for profiled in (True, False):
            frame(); torch.cuda.synchronize(dev)
            if profiled:
                _C.profile_reset(); _C.profile_enable(True)
            torch.cuda.synchronize(dev)
            last = time.perf_counter()
            for i in range(frames):
                frame()
                now = time.perf_counter(); sample((now - last) * 1e3); last = now
            torch.cuda.synchronize(dev)
            if profiled:
                _C.profile_read(); _C.profile_enable(False)
    else:
        for i in range(frames):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            frame()
            torch.cuda.synchronize(dev)
            sample((time.perf_counter() - t0) * 1e3)
# run-length encode on the controller's state (aim, floor, far tiles, back-off > 0); a run carries its frames' mean / max time
out, prev, n, first, acc, mx = [], None, 0, 0, 0.0, 0.0
def close():
    out.append({"from": first, "frames": n, "ms_mean": round(acc / n, 2), "ms_max": round(mx, 1), "aim": prev[0], "floor": prev[1], "far_tiles_last": prev[2], "backoff": prev[3]})
for i, r in enumerate(rows):
    key = (r[1], r[2], r[3], r[4] > 0)
    if key != prev:
        if prev is not None:
            close()
        prev, n, first, acc, mx = key, 0, i, 0.0, 0.0
    n += 1; acc += r[0]; mx = max(mx, r[0])
close()
ms = sorted(r[0] for r in rows)
print(json.dumps({"frames": len(rows), "P": P, "W": W, "H": H, "ms_p10_p50_p90": [ms[len(ms) // 10], ms[len(ms) // 2], ms[(9 * len(ms)) // 10]], "mean_ms": round(sum(ms) / len(ms), 3),
                  "runs": len(out), "frames_over_2.2_ms": sum(1 for m in ms if m > 2.2)}))
for o in out[:400]:
    print(json.dumps(o))
