"""The full-size frames whose reference outputs are pinned by hash (tests/golden/ref_hip_fullsize_sha256.json): BASELINE.json's headline
and configs 2-5 at their quoted sizes, on the SURVEY 8(d) synthetic recipe (wg_scenes).  Shared by the generator
(make_fullsize_ref_hashes.py: runs the reference's own kernels, oracle/_ref -ffp-contract=off) and by the test that holds the product to
the pins (tests/test_reference_golden.py) -- a full-size parity pin that needs no reference binary at test time."""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "wild-gaussians_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

PINS = os.path.join(HERE, "ref_hip_fullsize_sha256.json")

# name: (P, W, H, colours, camera index among the eight config-4 cameras or None = the base camera)
FRAMES = {
    "headline_1M_1080p_sh3": (1_000_000, 1920, 1080, "sh", None),
    "config2_500k_1080p_sh3": (500_000, 1920, 1080, "sh", None),
    "config3_3M_1600x1200_precomp": (3_000_000, 1600, 1200, "precomp", None),
    **{f"config4_camera{k}_1M_1080p_sh3": (1_000_000, 1920, 1080, "sh", k) for k in range(1, 8)},   # camera 0 is the headline frame
    "config5_10M_4K_sh3": (10_000_000, 3840, 2160, "sh", None),
}


def frame_inputs(name):
    """(cloud, cam, sh_degree) of a pinned frame."""
    import wg_scenes as S
    import wg_viewparallel as VP
    P, W, H, colours, k = FRAMES[name]
    cloud = S.make_cloud(P, W, H, sh_degree=3 if colours == "sh" else None, seed=0)
    cam = S.make_camera(W, H) if k is None else VP.view_cameras(8, W, H)[k]
    return cloud, cam, (3 if colours == "sh" else 0)


def sha(a, dtype):
    return hashlib.sha256(np.ascontiguousarray(np.asarray(a).reshape(-1), dtype=dtype).tobytes()).hexdigest()


def digest(num_rendered, radii, n_contrib, final_T):
    """What is pinned per frame: num_rendered, and the SHA-256 of radii (int32[P]), n_contrib (uint32[H*W]) and final_T (float32[H*W], its
    bits) -- every output of the forward pass that does not depend on a colour sum's rounding."""
    return {"num_rendered": int(num_rendered), "radii_sha256": sha(radii, np.int32), "n_contrib_sha256": sha(np.asarray(n_contrib).astype(np.int64), np.uint32),
            "final_T_sha256": sha(final_T, np.float32), "visible": int((np.asarray(radii) > 0).sum()),
            "n_contrib_sum": int(np.asarray(n_contrib).astype(np.int64).sum())}
