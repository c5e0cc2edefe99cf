#!/usr/bin/env python3
"""VERDICT r4 "next" item 3, the premise: could K8 blend on FAST values and re-run on the exact path only where the stop decision
T (1 - alpha) < 1e-4 (forward.cu:367-372) is within rounding distance of its threshold?  Counted on the CPU oracle (which records, per pixel,
frag_T = min over its chain of |test_T * 1e4 - 1|: the relative distance of the nearest stop test from 1e-4): how many pixels, and how many
16x16 tiles holding at least one, lie inside a band delta -- for the range of deltas a provable bound could take.  Also the pair counts
SURVEY 8(d)'s secondary ceiling is stated in (evaluated and blended (pixel, entry) pairs of the reference's K8 walk).
usage: python tests/tools/fragile_stop_premise.py [--gaussians P --width W --height H --colors sh|precomp --scale-mult S]   (CPU; ~1 min at 1 M / 1080p)"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "wild-gaussians_amd"))
import numpy as np  # noqa: E402
import wg_scenes as S  # noqa: E402
from oracle import oracle  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--gaussians", type=int, default=1_000_000)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--colors", default="sh")
ap.add_argument("--scale-mult", type=float, default=1.0)
a = ap.parse_args()
oracle.build()
W, H, P = a.width, a.height, a.gaussians
deg = 3 if a.colors == "sh" else None
cloud = S.make_cloud(P, W, H, sh_degree=deg, seed=0, scale_mult=a.scale_mult)
cam = S.make_camera(W, H)
t0 = time.time()
o = oracle.run_scene(cloud, cam, sh_degree=deg if deg is not None else 0)
ctx = o["ctx"]
fT = ctx.get("frag_T").astype(np.float64)
nev, nbl, ncon = ctx.get("n_evaluated").astype(np.int64), ctx.get("n_blended").astype(np.int64), ctx.get("n_contrib").astype(np.int64)
gx, gy = (W + 15) // 16, (H + 15) // 16
pad = np.full((gy * 16, gx * 16), np.inf)
pad[:H, :W] = fT
tile_min = pad.reshape(gy, 16, gx, 16).min(axis=(1, 3))
out = {"workload": f"{P} Gaussians, {W}x{H}, {a.colors}" + ("" if a.scale_mult == 1.0 else f", scales x{a.scale_mult:g}"), "oracle_seconds": round(time.time() - t0, 1),
       "num_rendered": int(o["num_rendered"]), "pixels": W * H, "tiles": gx * gy,
       "pairs": {"evaluated_by_the_reference_walk": int(nev.sum()), "blended": int(nbl.sum()), "sum_n_contrib": int(ncon.sum()),
                 "note": "evaluated = list entries a pixel's thread looks at before it stops (forward.cu:340-366); blended = the pairs that pass both skips "
                         "and the stop test, i.e. what K9 differentiates"},
       "pixels_that_reach_a_stop_test": int((fT < 1e29).sum()),
       "band": {}}
for d in (1e-7, 3e-7, 1e-6, 3e-6, 1e-5, 3e-5, 1e-4, 1e-3):
    px = int((fT < d).sum())
    tl = int((tile_min < d).sum())
    out["band"][f"{d:g}"] = {"pixels": px, "tiles": tl, "tile_fraction": round(tl / (gx * gy), 5)}
print(json.dumps(out, indent=1))
