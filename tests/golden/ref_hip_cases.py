"""The small cases run through the REAL reference (oracle/_ref, hipcc build of its CUDA sources) to make
tests/golden/ref_hip_golden.npz.  Inputs are stored in the fixture too, so the tests do not depend on this file."""
from __future__ import annotations

import numpy as np

W, H = 120, 72  # 8 x 5 tiles of 16 x 16, the right column and bottom row partial
P = 400


def _cov3d(scales, rotations, scale_modifier=1.0):
    """Upper triangle of R S S^T R^T in the reference's order (computeCov3D, forward.cu:117-149), float64 -> float32."""
    q = rotations.astype(np.float64)
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                  2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                  2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], axis=1).reshape(-1, 3, 3)
    S = scales.astype(np.float64) * scale_modifier
    M = R * S[:, None, :]
    Sig = M @ M.transpose(0, 2, 1)
    return np.stack([Sig[:, 0, 0], Sig[:, 0, 1], Sig[:, 0, 2], Sig[:, 1, 1], Sig[:, 1, 2], Sig[:, 2, 2]], axis=1).astype(np.float32)


def cases():
    """-> list of (name, cloud, cam, kwargs) with kwargs for run_scene (sh_degree, kernel_size, bg, scale_modifier,
    subpixel_offset) -- every argument of the operator takes a non-default value somewhere."""
    import wg_scenes as S
    out = []
    cam0 = S.make_camera(W, H)
    out.append(("sh3_defaults", S.make_cloud(P, W, H, sh_degree=3, seed=11, scale_mult=8.0), cam0, dict(sh_degree=3)))
    rng = np.random.default_rng(5)
    so = rng.uniform(-0.5, 0.5, size=(H, W, 2)).astype(np.float32)
    out.append(("sh1_kernel_bg_modifier_subpixel", S.make_cloud(P, W, H, sh_degree=1, seed=12, scale_mult=8.0), S.make_camera(W, H, yaw_deg=5.0),
                dict(sh_degree=1, kernel_size=0.3, bg=np.array([0.2, 0.5, 0.8], np.float32), scale_modifier=1.3, subpixel_offset=so)))
    out.append(("precomputed_colours_yaw", S.make_cloud(P, W, H, sh_degree=None, seed=13, scale_mult=10.0), S.make_camera(W, H, yaw_deg=-10.0),
                dict(sh_degree=0, kernel_size=0.0)))
    c = S.make_cloud(P, W, H, sh_degree=None, seed=14, scale_mult=8.0)
    c["cov3D_precomp"] = _cov3d(c.pop("scales"), c.pop("rotations"))
    out.append(("precomputed_covariances", c, cam0, dict(sh_degree=0)))
    c = S.make_cloud(P, W, H, sh_degree=0, seed=15, scale_mult=10.0)
    c["means3D"][:, 2] -= 2.5  # a third of the cloud behind the near plane / the camera
    c["means3D"][::7, 0] *= 3.0  # and some far outside the frustum sideways
    out.append(("sh0_near_plane_and_offscreen", c, cam0, dict(sh_degree=0)))
    out.append(("sh3_active_degree2_dense", S.make_cloud(P, W, H, sh_degree=3, seed=16, scale_mult=25.0), cam0, dict(sh_degree=2)))
    c = S.make_cloud(P, W, H, sh_degree=1, seed=17, scale_mult=12.0)
    c["opacities"][::3] = 0.999  # opaque fronts: the T < 1e-4 stop is reached
    c["opacities"][1::5] = 0.003  # below the 1/255 alpha cut everywhere
    out.append(("sh1_opaque_and_faint", c, cam0, dict(sh_degree=1, bg=np.array([1.0, 1.0, 1.0], np.float32))))
    return out
