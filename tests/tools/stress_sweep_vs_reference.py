#!/usr/bin/env python3
"""The argument sweep of tests/test_parity_gpu.py (cases beyond the 30 in CI) with the REFERENCE'S OWN KERNELS as the checker:
oracle/_ref (hipcc build of the reference's CUDA sources, -ffp-contract=off variant) and the product on the same GPU.
usage: python tests/tools/stress_sweep_vs_reference.py [first] [count]   -> one summary line (and one line per deviation)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "wild-gaussians_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import wg_scenes as S
from oracle.ref_hip import ref_hip
from tests.wg_testlib import run_hip, rel_err
from tests.test_parity_gpu import _sweep_case

first, count = int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 100
stats = dict(cases=0, radii_mismatch_cases=0, pixel_flip_cases=0, pixels_over=0, pixels=0, grad_over_cases=0, worst_grad=0.0, worst_fwd_p999=0.0)
for i in range(first, first + count):
    cloud, cam, deg, kw, W, H = _sweep_case(i)
    cot = S.make_cotangent(W, H, seed=3000 + i)
    r = ref_hip.run_scene(cloud, cam, sh_degree=deg, cotangent=cot, variant="nofma", **kw)
    h = run_hip(cloud, cam, sh_degree=deg, cotangent=cot, **kw)
    err = np.abs(h["color"].astype(np.float64) - r["color"]).max(axis=0)
    over = int((err > 1e-4).sum())
    g = {k: rel_err(v.reshape(r["grads"][k].shape), r["grads"][k]) for k, v in h["grads"].items()}
    stats["cases"] += 1
    stats["pixels"] += err.size
    stats["pixels_over"] += over
    if (h["radii"] != r["radii"]).any():
        stats["radii_mismatch_cases"] += 1
        print("radii differ: case", i, int((h["radii"] != r["radii"]).sum()))
    if over:
        stats["pixel_flip_cases"] += 1
    else:
        stats["worst_grad"] = max(stats["worst_grad"], max(g.values()))
        stats["worst_fwd_p999"] = max(stats["worst_fwd_p999"], float(err.max()))
    if max(g.values()) > 1e-3:
        stats["grad_over_cases"] += 1
        print("gradient over 1e-3: case", i, "pixels over 1e-4:", over, "worst", max(g, key=g.get), max(g.values()))
print(f"cases {first}..{first + count - 1} vs the reference's own kernels:", stats)
