#!/usr/bin/env python3
"""How evenly do the forward render kernel's waves load the SIMDs?  Needs a library built with -DWG_FWD_PROBE=1 for render_fwd.hip
(git apply experiments/r3_forward_probe.patch; scripts/ab_variants.sh probe "render_fwd.hip:-DWG_FWD_PROBE=1"; WG_RASTERIZER_LIB points
at it; git apply -R ... afterwards: the probe is not part of the product source): every wave records its start / end
on the 100 MHz real-time counter and the SIMD it ran on.  One wave per tile, all ~8 k waves resident at once (8 per SIMD): the kernel
ends when the SIMD with the largest SUM of tile costs ends.

usage: WG_RASTERIZER_LIB=.../build/probe/libwg_rasterizer.so python scripts/probe_forward_balance.py [gaussians width height]"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wild-gaussians_amd"))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import wg_scenes as S  # noqa: E402
from diff_gaussian_rasterization import GaussianRasterizer, _C  # noqa: E402
from tests.wg_testlib import make_settings, to_dev  # noqa: E402

P, W, H = (int(a) for a in (sys.argv[1:4] + ["1000000", "1920", "1080"][len(sys.argv) - 1:]))
dev = torch.device("cuda", 0)
cloud = S.make_cloud(P, W, H, sh_degree=3, seed=0)
rast = GaussianRasterizer(make_settings(S.make_camera(W, H), 3, device=dev))
t = {k: to_dev(v, dev) for k, v in cloud.items()}
m2 = torch.zeros_like(t["means3D"])
for _ in range(5):
    with torch.no_grad():
        rast(means3D=t["means3D"], means2D=m2, opacities=t["opacities"], shs=t["shs"], scales=t["scales"], rotations=t["rotations"])
torch.cuda.synchronize()
tiles = ((W + 15) // 16) * ((H + 15) // 16)
buf = np.zeros(4 * 65536, np.uint64)
lib = _C._lib
lib.wg_probe_fetch.restype, lib.wg_probe_fetch.argtypes = C.c_int, [C.c_void_p, C.c_size_t]
assert lib.wg_probe_fetch(buf.ctypes.data, buf.nbytes) == 0
p = buf.reshape(-1, 4)[:tiles]
t0, t1, hw = p[:, 0].astype(np.int64), p[:, 1].astype(np.int64), p[:, 2]
hwid, xcc = (hw & 0xffffffff).astype(np.int64), (hw >> 32).astype(np.int64) & 0xf
simd = (xcc << 20) | (hwid & 0xfff0)          # XCC, SE / SH / CU / pipe / SIMD fields of HW_ID (its wave-slot bits [3:0] dropped)
dur = (t1 - t0) / 100.0                       # us
span = (t1.max() - t0.min()) / 100.0
ids, inv = np.unique(simd, return_inverse=True)
per_simd_sum = np.bincount(inv, weights=dur)
per_simd_end = np.array([(t1[inv == i].max() - t0.min()) / 100.0 for i in range(len(ids))])
per_simd_n = np.bincount(inv)
q = lambda v, f: round(float(np.quantile(v, f)), 2)  # noqa: E731
print(json.dumps({
    "workload": f"{P} Gaussians, {W}x{H}", "tiles": int(tiles), "simds_seen": int(len(ids)), "kernel_span_us": round(span, 2),
    "wave_duration_us": {"mean": round(float(dur.mean()), 2), "p10": q(dur, 0.1), "p50": q(dur, 0.5), "p90": q(dur, 0.9), "max": q(dur, 1.0)},
    "waves_per_simd": {"min": int(per_simd_n.min()), "p50": int(np.median(per_simd_n)), "max": int(per_simd_n.max())},
    "simd_finish_time_us": {"p10": q(per_simd_end, 0.1), "p50": q(per_simd_end, 0.5), "p90": q(per_simd_end, 0.9), "max": q(per_simd_end, 1.0)},
    "mean_over_max_simd_finish": round(float(per_simd_end.mean() / per_simd_end.max()), 3),
    "latest_wave_start_us": round(float((t0.max() - t0.min()) / 100.0), 2),
}))
