"""Helpers shared by the GPU parity tests, __graft_entry__.smoke() and bench.py's correctness line:
run the HIP path (through the drop-in package -> C-ABI) and the CPU oracle on the same wg_scenes inputs."""
from __future__ import annotations

import numpy as np
import torch

GRAD_NAMES = ["means3D", "means2D", "sh", "colors_precomp", "opacities", "scales", "rotations", "cov3Ds_precomp"]


def to_dev(a, device="cuda"):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


def make_settings(cam, sh_degree, kernel_size=0.1, bg=None, subpixel_offset=None, scale_modifier=1.0, debug=False,
                  return_accumulation=True, device="cuda"):
    from diff_gaussian_rasterization import GaussianRasterizationSettings
    H, W = cam["height"], cam["width"]
    bg_t = torch.zeros(3, device=device) if bg is None else to_dev(np.asarray(bg, np.float32), device)
    so = torch.zeros((H, W, 2), device=device) if subpixel_offset is None else to_dev(subpixel_offset.astype(np.float32), device)
    return GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], kernel_size=kernel_size,
        subpixel_offset=so, bg=bg_t, scale_modifier=scale_modifier, viewmatrix=to_dev(cam["viewmatrix"], device),
        projmatrix=to_dev(cam["projmatrix"], device), sh_degree=sh_degree, campos=to_dev(cam["campos"], device),
        prefiltered=False, debug=debug, return_accumulation=return_accumulation)


def run_hip(cloud, cam, sh_degree=3, kernel_size=0.1, bg=None, subpixel_offset=None, cotangent=None, scale_modifier=1.0,
            device="cuda", binning_capacity=None):
    """Forward (+ backward when a cotangent is given) through GaussianRasterizer; numpy results.  binning_capacity: the
    fixed-capacity forward (wg_forward_args::binning_capacity)."""
    from diff_gaussian_rasterization import GaussianRasterizer
    rs = make_settings(cam, sh_degree, kernel_size, bg, subpixel_offset, scale_modifier, device=device)
    t = {k: to_dev(v, device).requires_grad_(cotangent is not None) for k, v in cloud.items()}
    means2D = torch.zeros_like(t["means3D"], requires_grad=cotangent is not None)
    rast = GaussianRasterizer(rs)
    color, radii, acc = rast(means3D=t["means3D"], means2D=means2D, opacities=t["opacities"], shs=t.get("shs"),
                             colors_precomp=t.get("colors_precomp"), scales=t.get("scales"), rotations=t.get("rotations"),
                             cov3D_precomp=t.get("cov3D_precomp"), **({} if binning_capacity is None else {"binning_capacity": binning_capacity}))
    out = dict(color=color.detach().cpu().numpy(), radii=radii.cpu().numpy(), accumulation=acc.detach().cpu().numpy())
    if cotangent is not None:
        color.backward(to_dev(cotangent, device))
        g = dict(means3D=t["means3D"].grad, means2D=means2D.grad, opacities=t["opacities"].grad)
        for k_in, k_out in (("shs", "sh"), ("colors_precomp", "colors_precomp"), ("scales", "scales"),
                            ("rotations", "rotations"), ("cov3D_precomp", "cov3Ds_precomp")):
            if k_in in t:
                g[k_out] = t[k_in].grad
        out["grads"] = {k: v.detach().cpu().numpy() for k, v in g.items() if v is not None}
    return out


def run_hip_native(cloud, cam, sh_degree=3, kernel_size=0.1, bg=None, subpixel_offset=None, device="cuda"):
    """Forward through the native module (_C -> C-ABI) keeping the scratch buffers; returns typed views."""
    from diff_gaussian_rasterization import _C
    rs = make_settings(cam, sh_degree, kernel_size, bg, subpixel_offset, device=device)
    e = torch.Tensor([])
    t = {k: to_dev(v, device) for k, v in cloud.items()}
    R, color, radii, gb, bb, ib = _C.rasterize_gaussians(
        rs.bg, t["means3D"], t.get("colors_precomp", e), t["opacities"], t.get("scales", e), t.get("rotations", e), 1.0,
        t.get("cov3D_precomp", e), rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.kernel_size, rs.subpixel_offset,
        rs.image_height, rs.image_width, t.get("shs", e), sh_degree, rs.campos, False, False)
    P = t["means3D"].shape[0]
    views = dict(geometry=_C.view_geometry(gb, P), binning=_C.view_binning(bb, R), image=_C.view_image(ib, rs.image_height, rs.image_width))
    return dict(num_rendered=R, color=color, radii=radii, buffers=(gb, bb, ib), views=views)


def rel_err(a, ref):
    """max|g - g_ref| / (max|g_ref| + 1e-12)."""
    a, ref = np.asarray(a, np.float64), np.asarray(ref, np.float64)
    return float(np.abs(a - ref).max() / (np.abs(ref).max() + 1e-12)) if ref.size else 0.0


# thresholds on the oracle's per-pixel distance-to-decision maps below which a pixel's outcome legitimately
# depends on sub-ulp differences between two correct float32 evaluations (see tests/test_parity_gpu.py)
FRAG_ALPHA = 2e-5
FRAG_T = 2e-4


def compare_forward(hip_color, oracle_out, atol=1e-4):
    """-> dict(max_err_solid, n_fragile, n_bad_fragile, max_err_all).  'solid' pixels are those whose every
    threshold decision in the oracle has a safety margin; they must agree to `atol`."""
    ctx = oracle_out["ctx"]
    err = np.abs(hip_color.astype(np.float64) - oracle_out["color"].astype(np.float64)).max(axis=0)
    fragile = (ctx.get("frag_alpha") < FRAG_ALPHA) | (ctx.get("frag_T") < FRAG_T)
    solid = ~fragile
    return dict(max_err_solid=float(err[solid].max()) if solid.any() else 0.0, max_err_all=float(err.max()),
                n_fragile=int(fragile.sum()), n_pixels=int(err.size), n_over_in_fragile=int((err[fragile] > atol).sum()),
                solid_mask=solid)


def compare_grads(hip_grads, oracle_grads):
    res = {}
    for k in GRAD_NAMES:
        if k in hip_grads and k in oracle_grads and oracle_grads[k].size:
            res[k] = rel_err(hip_grads[k].reshape(oracle_grads[k].shape), oracle_grads[k])
    return res


def _S():
    import wg_scenes
    return wg_scenes


def sweep_case(i):
    """Case i of a deterministic sweep: odd frame sizes, fields of view, rotated / translated cameras, every SH degree and
    precomputed colours, mip-filter sizes, non-zero background, sub-pixel offsets, scale_modifier, precomputed covariances."""
    rng = np.random.default_rng(1000 + i)
    W = int(rng.integers(17, 230))
    H = int(rng.integers(17, 170))
    fov = float(rng.uniform(35.0, 95.0))
    yaw = float(rng.uniform(-12.0, 12.0))
    cam = _S().make_camera(W, H, fov_x_deg=fov, yaw_deg=yaw)
    deg = [None, 0, 1, 2, 3][i % 5]
    P = int(rng.integers(50, 2500))
    cloud = _S().make_cloud(P, W, H, sh_degree=deg, seed=2000 + i, fov_x_deg=fov, scale_mult=float(rng.uniform(0.5, 9.0)))
    if i % 7 == 3:   # some Gaussians behind / too close to the camera, some far outside the frame
        cloud["means3D"][: P // 5, 2] = rng.uniform(-1.0, 0.25, size=P // 5).astype(np.float32)
        cloud["means3D"][P // 5: P // 4, 0] *= 6.0
    kw = dict(kernel_size=float(rng.choice([0.0, 0.1, 0.3, 1.0])),
              bg=rng.uniform(0, 1, size=3).astype(np.float32) if i % 2 else None,
              subpixel_offset=rng.uniform(-0.5, 0.5, size=(H, W, 2)).astype(np.float32) if i % 3 == 0 else None,
              scale_modifier=float(rng.choice([1.0, 0.6, 1.7])))
    if i % 6 == 5:   # covariances instead of scales + rotations (the operator accepts exactly one of the two)
        s, q = cloud.pop("scales").astype(np.float64), cloud.pop("rotations").astype(np.float64)
        r, x, y, z = q.T
        Rm = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                       2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                       2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], axis=1).reshape(-1, 3, 3)
        M = Rm * (s * kw["scale_modifier"])[:, None, :]
        Sg = M @ M.transpose(0, 2, 1)
        cloud["cov3D_precomp"] = np.stack([Sg[:, 0, 0], Sg[:, 0, 1], Sg[:, 0, 2], Sg[:, 1, 1], Sg[:, 1, 2], Sg[:, 2, 2]], 1).astype(np.float32)
    return cloud, cam, (deg if deg is not None else 0), kw, W, H
