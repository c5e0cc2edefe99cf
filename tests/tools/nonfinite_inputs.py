#!/usr/bin/env python3
"""Non-finite and degenerate GEOMETRY inputs (NaN / +-Inf / zero / negative in means, scales, rotations, opacities): what the reference's
own kernels (oracle/_ref, -ffp-contract=off build) make of them, and what the product makes of them -- per category, forward only:
radii, num_rendered, which pixels are non-finite, and the finite pixels' values.  Colours stay finite.  Exploration tool (GPU).
usage: python tests/tools/nonfinite_inputs.py [P W H per_category]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "wild-gaussians_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import wg_scenes as S
from oracle.ref_hip import ref_hip
from tests.wg_testlib import run_hip

P, W, H, K = (int(a) for a in (sys.argv[1:5] + ["20000", "640", "360", "8"][len(sys.argv[1:5]):]))
SIDES = os.environ.get("WG_SIDES", "both")   # "ours" / "ref": run one side only (a device fault ends the process: find out whose it is)
import ref_mode_checks as RC

CATS = RC.NONFINITE_CATEGORIES
cam = S.make_camera(W, H)
base = S.make_cloud(P, W, H, sh_degree=1, seed=11, scale_mult=2.0)
rng = np.random.default_rng(3)


def compare(name, cloud):
    import torch
    print(f"{name:16s} ...", end="", flush=True)
    if SIDES != "both":
        o = run_hip(cloud, cam, sh_degree=1) if SIDES == "ours" else ref_hip.run_scene(cloud, cam, sh_degree=1, variant="nofma")
        torch.cuda.synchronize()
        print(f" {SIDES}: ok, non-finite pixels {int((~np.isfinite(o['color']).all(axis=0)).sum())}, radii > 0: {int((o['radii'] > 0).sum())}", flush=True)
        return
    r = ref_hip.run_scene(cloud, cam, sh_degree=1, variant="nofma")
    h = run_hip(cloud, cam, sh_degree=1)
    rn, hn = ~np.isfinite(r["color"]).all(axis=0), ~np.isfinite(h["color"]).all(axis=0)
    both = ~rn & ~hn
    d = np.abs(h["color"].astype(np.float64) - r["color"])[:, both]
    acc_bad = int((~np.isfinite(h["accumulation"])).sum())
    print(f" radii differ {int((h['radii'] != r['radii']).sum()):5d}   non-finite pixels ref {int(rn.sum()):7d} ours {int(hn.sum()):7d} (mask differs {int((rn != hn).sum())})"
          f"   finite pixels: max diff {float(d.max()) if d.size else 0.0:.2e}, over 1e-4: {int((d.max(axis=0) > 1e-4).sum()) if d.size else 0}"
          f"   accumulation differs {int((np.nan_to_num(h['accumulation'], nan=-7.0) != np.nan_to_num(r['accumulation'], nan=-7.0)).sum())}")


compare("clean", base)
every = {k: v.copy() for k, v in base.items()}
for name in CATS:
    ids = rng.choice(P, size=K, replace=False)
    every = RC.poison(every, name, ids)
    compare(name, RC.poison(base, name, ids))
if SIDES == "ours":   # (the reference's own kernels end in a memory access fault on the combined cloud: profiles/r5/nonfinite_inputs_ref.log)
    compare("all of the above", every)
