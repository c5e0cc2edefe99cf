#!/usr/bin/env python3
"""One case of the argument sweep against the float32 AND the float64 oracle (is a deviation a threshold flip of a fragile pixel?).
usage: python tests/tools/debug_sweep_case.py <case>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "wild-gaussians_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import wg_scenes as S
from oracle import oracle
from tests.wg_testlib import run_hip, compare_forward, compare_grads
from tests.test_parity_gpu import _sweep_case
i = int(sys.argv[1])
cloud, cam, deg, kw, W, H = _sweep_case(i)
cot = S.make_cotangent(W, H, seed=3000 + i)
o32 = oracle.run_scene(cloud, cam, sh_degree=deg, cotangent=cot, **kw)
kw64 = {k: (v.astype(np.float64) if isinstance(v, np.ndarray) else v) for k, v in kw.items()}
c64 = {k: v.astype(np.float64) for k, v in cloud.items()}
cam64 = {k: (v.astype(np.float64) if isinstance(v, np.ndarray) else v) for k, v in cam.items()}
o64 = oracle.run_scene(c64, cam64, sh_degree=deg, cotangent=cot.astype(np.float64), precision="f64", **kw64)
h = run_hip(cloud, cam, sh_degree=deg, cotangent=cot, **kw)
print("kw", {k: (v if not isinstance(v, np.ndarray) else v.shape) for k, v in kw.items()}, "deg", deg)
print("hip vs f32 oracle", compare_grads(h["grads"], o32["grads"]))
print("hip vs f64 oracle", compare_grads(h["grads"], o64["grads"]))
print("f32 oracle vs f64 oracle", compare_grads(o32["grads"], o64["grads"]))
c = compare_forward(h["color"], o32)
print({k: v for k, v in c.items() if k != "solid_mask"})
nc32, nc64 = o32["ctx"].get("n_contrib"), o64["ctx"].get("n_contrib")
print("n_contrib differs f32 vs f64 at", int((nc32 != nc64).sum()), "pixels")
