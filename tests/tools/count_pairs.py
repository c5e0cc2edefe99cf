#!/usr/bin/env python3
"""The inputs of SURVEY 8(d)'s secondary (compute) ceiling of the two render kernels, COUNTED (VERDICT r4 "next" item 2): one forward +
backward pass of a bench workload through the counting VARIANT build of the library (-DWG_COUNT_PAIRS=1: wild-gaussians_amd/build/count/,
never the product library), whose K8 / K9 add up what they do; beside them the pair counts of the REFERENCE's walk from the CPU oracle.
Build the variant where hipcc is:   WG_BUILD_VARIANT=count WG_EXTRA_FLAGS=-DWG_COUNT_PAIRS=1 python wild-gaussians_amd/build.py
Run on the GPU box:                 python tests/tools/count_pairs.py [--gaussians P --width W --height H --colors sh|precomp --forward-only]
-> one JSON object (kept as profiles/pair_counts*.json, which bench.py reads for `roofline.compute`)."""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LIB = os.path.join(ROOT, "wild-gaussians_amd", "build", "count", "libwg_rasterizer.so")
if os.environ.get("WG_RASTERIZER_LIB") != LIB:   # the binding reads the variable at import: start over with it set
    assert os.path.exists(LIB), "no counting build: " + LIB
    os.execve(sys.executable, [sys.executable] + sys.argv, dict(os.environ, WG_RASTERIZER_LIB=LIB, WG_BINDING="ctypes"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "wild-gaussians_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import wg_scenes as S  # noqa: E402
from tests.wg_testlib import run_hip  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--gaussians", type=int, default=1_000_000)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--colors", default="sh")
ap.add_argument("--scale-mult", type=float, default=1.0)
ap.add_argument("--forward-only", action="store_true")
ap.add_argument("--oracle-gaussians", type=int, default=0, help="0 = the whole cloud; the CPU oracle's key sort is single-threaded")
a = ap.parse_args()
W, H, P = a.width, a.height, a.gaussians
deg = 3 if a.colors == "sh" else None
d = deg if deg is not None else 0
cloud = S.make_cloud(P, W, H, sh_degree=deg, seed=0, scale_mult=a.scale_mult)
cam = S.make_camera(W, H)
lib = C.CDLL(LIB)
u8 = (C.c_ulonglong * 8)()
lib.wg_debug_fwd_counters(u8, 1)
lib.wg_debug_bwd_counters((C.c_ulonglong * 16)(), 1)
h = run_hip(cloud, cam, sh_degree=d, cotangent=None if a.forward_only else S.make_cotangent(W, H))
lib.wg_debug_fwd_counters(u8, 1)
fwd = list(u8)[:6]
u16 = (C.c_ulonglong * 16)()
lib.wg_debug_bwd_counters(u16, 1)
bwd = list(u16)[:14]
import bench  # noqa: E402
out = {"workload": f"{P} Gaussians, {W}x{H}, {a.colors}" + ("" if a.scale_mult == 1.0 else f", scales x{a.scale_mult:g}"),
       **{k: v for k, v in bench.profile_stamps().items() if k.endswith("_sha")}, "collected": time.strftime("%Y-%m-%d"), "device": torch.cuda.get_device_name(0),
       "render_forward": {"instances_visited": fwd[0], "strip_evaluations": fwd[1], "pairs_evaluated": 64 * fwd[1], "pairs_evaluated_on_accumulating_pixels": fwd[2],
                          "pairs_passing_both_skips": fwd[3], "pixels_stopped": fwd[4], "pairs_blended": fwd[3] - fwd[4], "strip_evaluations_without_a_passing_pair": fwd[5]},
       "render_backward": None if a.forward_only else {"instances_visited": bwd[0], "strip_evaluations": bwd[1], "pairs_evaluated": 64 * bwd[1],
                                                       "pairs_at_or_before_the_last_contributor": bwd[2], "pairs_contributing": bwd[3],
                                                       "instances_reduced": bwd[4], "strip_evaluations_without_a_contributing_pair": bwd[5],
                                                       "instances_reduced_by_contributing_lanes": {"1": bwd[6], "2-4": bwd[7], "5-16": bwd[8], "17-64": bwd[9]},
                                                       "strip_evaluations_by_contributing_lanes": {"1-8": bwd[10], "9-24": bwd[11], "25-48": bwd[12], "49-64": bwd[13]}},
       "what": "per launch; a strip evaluation is one wave-wide evaluation of an instance on an 8x8 strip = 64 (pixel, entry) pairs"}
# the reference's walk on the same frame (forward.cu:340-381: every pixel looks at every entry of its tile's list until it stops)
from oracle import oracle  # noqa: E402
oracle.build()
Po = a.oracle_gaussians or P
sub = cloud if Po >= P else {k: np.ascontiguousarray(v[:Po]) for k, v in cloud.items()}
o = oracle.run_scene(sub, cam, sh_degree=d)
ctx = o["ctx"]
out["reference_walk"] = {"gaussians": Po, "pairs_evaluated": int(ctx.get("n_evaluated").astype(np.int64).sum()),
                         "pairs_blended": int(ctx.get("n_blended").astype(np.int64).sum()), "num_rendered": int(o["num_rendered"])}
if Po >= P:
    assert out["reference_walk"]["pairs_blended"] == out["render_forward"]["pairs_blended"] or abs(
        out["reference_walk"]["pairs_blended"] - out["render_forward"]["pairs_blended"]) < 1e-5 * out["reference_walk"]["pairs_blended"], out
print(json.dumps(out))
