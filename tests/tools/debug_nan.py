#!/usr/bin/env python3
"""Exploration (GPU): what the reference's kernels and the product do with a NaN conic, per poisoned Gaussian (tests/tools/nonfinite_inputs.py's
'rot NaN' / 'scale NaN' categories): the product's splat record, the tile, and ref / product n_contrib and colour inside that tile."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "wild-gaussians_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import wg_scenes as S
from oracle.ref_hip import ref_hip
from tests.wg_testlib import run_hip_native
import ref_mode_checks as RC

P, W, H, K = 20000, 640, 360, 8
cam = S.make_camera(W, H)
base = S.make_cloud(P, W, H, sh_degree=1, seed=11, scale_mult=2.0)
rng = np.random.default_rng(3)
for name in RC.NONFINITE_CATEGORIES:
    ids = rng.choice(P, size=K, replace=False)
    if name not in sys.argv[1:]:
        continue
    cloud = RC.poison(base, name, ids)
    r = ref_hip.run_scene(cloud, cam, sh_degree=1, variant="nofma")
    n = run_hip_native(cloud, cam, sh_degree=1)
    g = n["views"]["geometry"]
    im = n["views"]["image"]
    rec = g["splats"].cpu().numpy().reshape(P, 12)
    tt = g["tiles_touched"].cpu().numpy()
    nc = im["n_contrib"].cpu().numpy().reshape(H, W)
    col = n["color"].cpu().numpy()
    print("==", name, "num_rendered ours/ref", n["num_rendered"], r["num_rendered"])
    for i in ids:
        mx, my = rec[i, 0], rec[i, 1]
        tx, ty = int(mx) // 16, int(my) // 16
        print(f" id {i} radius ours {int(n['radii'][i])} ref {int(r['radii'][i])} tiles_touched {tt[i]} rec {rec[i, :8]} tile ({tx},{ty})")
        if not (0 <= tx < (W + 15) // 16 and 0 <= ty < (H + 15) // 16):
            continue
        ys, xs = slice(16 * ty, min(H, 16 * ty + 16)), slice(16 * tx, min(W, 16 * tx + 16))
        d = np.abs(col[:, ys, xs].astype(np.float64) - r["color"][:, ys, xs]).max(axis=0)
        print(f"   tile pixels differing > 1e-4: {int((d > 1e-4).sum())} of {d.size}; n_contrib equal: {int((nc[ys, xs].astype(np.int64) == r['n_contrib'][ys, xs].astype(np.int64)).sum())}")
        print("   ours n_contrib row0", nc[ys, xs][0, :8], " ref", r["n_contrib"][ys, xs][0, :8].astype(np.int64))
        print("   ours col row0 ch0", col[0, ys, xs][0, :4], " ref", r["color"][0, ys, xs][0, :4])
        fT = im["final_T"].cpu().numpy().reshape(H, W)
        print("   ours final_T row0", fT[ys, xs][0, :6], " ref", r["final_T"][ys, xs][0, :6])
        rg = im["ranges"].cpu().numpy().reshape(-1, 2)
        tile = ty * ((W + 15) // 16) + tx
        pl = n["views"]["binning"]["point_list"].cpu().numpy()[rg[tile, 0]:rg[tile, 1]]
        dep = g["depths"].cpu().numpy()
        pos = np.nonzero(pl == i)[0]
        print(f"   tile list length {len(pl)}, position of the poisoned id (1-based) {pos + 1}, its depth {dep[i]}, depths around it {dep[pl[max(0, int(pos[0]) - 2):int(pos[0]) + 3]] if len(pos) else None}")
        print("   ours n_contrib min/max in tile", nc[ys, xs].min(), nc[ys, xs].max(), " ref", r["n_contrib"][ys, xs].min(), r["n_contrib"][ys, xs].max())
    dd = np.abs(col.astype(np.float64) - r["color"]).max(axis=0)
    yy, xx = np.nonzero(dd > 1e-4)
    print("  differing pixels total", yy.size, "tiles:", sorted(set(zip((xx // 16).tolist(), (yy // 16).tolist())))[:40])
