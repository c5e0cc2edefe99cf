#!/usr/bin/env python3
"""How evenly do the two render kernels' waves load the 1 024 SIMDs?  Needs the probe VARIANT build of the library
(scripts/ab_variants.sh probe "render_fwd.hip:-DWG_PROBE=1;render_bwd.hip:-DWG_PROBE=1"): every wave records its start / end on the
100 MHz real-time counter, the SIMD it ran on and its tile.  One wave per tile; the forward kernel's ~8 k waves are resident at once
(8 per SIMD), the backward kernel's 6 per SIMD: a kernel ends when its most loaded SIMD ends.

usage: python scripts/probe_balance.py [gaussians width height] [--dump file.npz] [--lib path] [--option name=value ...]"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = [a for a in sys.argv[1:]]
dump = lib_path = None
if "--dump" in args:
    i = args.index("--dump"); dump = args[i + 1]; del args[i:i + 2]
if "--lib" in args:
    i = args.index("--lib"); lib_path = args[i + 1]; del args[i:i + 2]
options = []
while "--option" in args:
    i = args.index("--option"); options.append(args[i + 1]); del args[i:i + 2]
LIB = lib_path or os.path.join(ROOT, "wild-gaussians_amd", "build", "probe", "libwg_rasterizer.so")
if os.environ.get("WG_RASTERIZER_LIB") != LIB:   # the binding reads the variable at import: start over with it set
    assert os.path.exists(LIB), "no probe build: " + LIB
    os.execve(sys.executable, [sys.executable] + sys.argv, dict(os.environ, WG_RASTERIZER_LIB=LIB))
sys.path.insert(0, os.path.join(ROOT, "wild-gaussians_amd"))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import wg_scenes as S  # noqa: E402
from diff_gaussian_rasterization import GaussianRasterizer, _C  # noqa: E402
from tests.wg_testlib import make_settings, to_dev  # noqa: E402

P, W, H = (int(a) for a in (args[:3] + ["1000000", "1920", "1080"][len(args):]))
for kv in options:
    _C.set_option(kv.split("=")[0], int(kv.split("=")[1]))
dev = torch.device("cuda", 0)
cloud = S.make_cloud(P, W, H, sh_degree=3, seed=0)
rast = GaussianRasterizer(make_settings(S.make_camera(W, H), 3, device=dev))
t = {k: to_dev(v, dev).requires_grad_(True) for k, v in cloud.items()}
cot = torch.from_numpy(S.make_cotangent(W, H)).to(dev)
for _ in range(6):
    m2 = torch.zeros_like(t["means3D"], requires_grad=True)
    color = rast(means3D=t["means3D"], means2D=m2, opacities=t["opacities"], shs=t["shs"], scales=t["scales"], rotations=t["rotations"])[0]
    color.backward(cot)
torch.cuda.synchronize()
tiles = ((W + 15) // 16) * ((H + 15) // 16)
lib = C.CDLL(LIB)
out = {"workload": f"{P} Gaussians, {W}x{H}", "tiles": int(tiles), "options": options}
raw = {}
for name, fn in (("render_forward", "wg_probe_fetch"), ("render_backward", "wg_probe_fetch_bwd")):
    buf = np.zeros(4 * 65536, np.uint64)
    f = getattr(lib, fn)
    f.restype, f.argtypes = C.c_int, [C.c_void_p, C.c_size_t]
    assert f(buf.ctypes.data, buf.nbytes) == 0
    p = buf.reshape(-1, 4)[:tiles]
    raw[name] = p.copy()
    t0, t1, hw = p[:, 0].astype(np.int64), p[:, 1].astype(np.int64), p[:, 2]
    hwid, xcc = (hw & 0xffffffff).astype(np.int64), (hw >> 32).astype(np.int64) & 0xf
    simd = (xcc << 20) | (hwid & 0xfff0)          # XCC, SE / SH / CU / pipe / SIMD fields of HW_ID (its wave-slot bits [3:0] dropped)
    dur = (t1 - t0) / 100.0                       # us
    span = (t1.max() - t0.min()) / 100.0
    ids, inv = np.unique(simd, return_inverse=True)
    per_simd_end = np.array([(t1[inv == i].max() - t0.min()) / 100.0 for i in range(len(ids))])
    per_simd_n = np.bincount(inv)
    q = lambda v, f: round(float(np.quantile(v, f)), 2)  # noqa: E731
    out[name] = {
        "simds_seen": int(len(ids)), "kernel_span_us": round(span, 2),
        "wave_duration_us": {"mean": round(float(dur.mean()), 2), "p10": q(dur, 0.1), "p50": q(dur, 0.5), "p90": q(dur, 0.9), "max": q(dur, 1.0)},
        "waves_per_simd": {"min": int(per_simd_n.min()), "p50": int(np.median(per_simd_n)), "max": int(per_simd_n.max())},
        "simd_finish_time_us": {"p10": q(per_simd_end, 0.1), "p50": q(per_simd_end, 0.5), "p90": q(per_simd_end, 0.9), "max": q(per_simd_end, 1.0)},
        "mean_over_max_simd_finish": round(float(per_simd_end.mean() / per_simd_end.max()), 3),
        "latest_wave_start_us": round(float((t0.max() - t0.min()) / 100.0), 2),
    }
if dump:   # the frame's per-tile walked lengths beside the probes (the cost the launch orders are built from)
    from tests.wg_testlib import run_hip_native
    nat = run_hip_native(cloud, S.make_camera(W, H), sh_degree=3, device=dev)
    raw["tile_last"] = nat["views"]["image"]["tile_last"].cpu().numpy()
    raw["ranges"] = nat["views"]["image"]["ranges"].cpu().numpy()
print(json.dumps(out))
if dump:
    np.savez_compressed(dump, **raw)
