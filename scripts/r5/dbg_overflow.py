import sys, os
sys.path.insert(0, os.path.join(os.getcwd(), "wild-gaussians_amd")); sys.path.insert(0, os.getcwd())
import numpy as np, torch
import wg_scenes as S
from tests.wg_testlib import run_hip_native, run_hip
from diff_gaussian_rasterization import _C
W, H = 1920, 1080
cam = S.make_camera(W, H)
print("default speculative_forward", _C.get_option("speculative_forward"), flush=True)
small = S.make_cloud(20_000, W, H, sh_degree=None, seed=4)
alone = run_hip(small, cam, sh_degree=0)
for P in (300_000,):
    big = S.make_cloud(P, W, H, sh_degree=None, seed=3)
    big["scales"][:] = 50.0
    for fn in ("run_hip", "native", "run_hip"):
        try:
            if fn == "native":
                r = run_hip_native(big, cam, sh_degree=0); print(fn, "num_rendered", r["num_rendered"], flush=True)
            else:
                r = run_hip(big, cam, sh_degree=0); print(fn, "returned; radii>0", int((r["radii"] > 0).sum()), "color finite", bool(np.isfinite(r["color"]).all()), "color max", float(np.nanmax(r["color"])), flush=True)
        except Exception as ex:
            print(fn, "raised", repr(ex)[:200], flush=True)
