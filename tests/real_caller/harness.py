"""Run the reference's REAL training step -- `WildGaussians.train_iteration` (wildgaussians/method.py:1880-2024), unchanged --
on this repo's drop-in `diff_gaussian_rasterization` and `simple_knn` (BASELINE config 3).

TEST INFRASTRUCTURE.  The caller's modules come from tests/real_caller/_staged/ (byte-identical copies made by
stage_reference_caller.py, verified against manifest.json); `omegaconf` / `plyfile` are test-only stand-ins under shims/;
the `Dataset` (types.py:264-284) is synthetic; `uncertainty_mode=disabled` (config.py:77: the DINOv2 weights need network).
Everything the step executes besides those stand-ins is the reference's own Python.
"""
from __future__ import annotations

import importlib
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
STAGED = os.path.join(HERE, "_staged")
SHIMS = os.path.join(HERE, "shims")
PKG = os.path.join(ROOT, "wild-gaussians_amd")


def staged_available() -> bool:
    sys.path.insert(0, HERE)
    try:
        import stage_reference_caller as st
        return st.staged_ok()
    finally:
        sys.path.remove(HERE)


def import_method():
    """`import wildgaussians.method` with: the staged caller, the stand-ins, and THIS repo's operator packages on sys.path."""
    for p in (STAGED, SHIMS, PKG, ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    m = importlib.import_module("wildgaussians.method")
    import diff_gaussian_rasterization as dgr
    import simple_knn._C as knn
    assert m.GaussianRasterizer is dgr.GaussianRasterizer and m.distCUDA2 is knn.distCUDA2, "caller bound to another operator"
    assert os.path.abspath(m.__file__).startswith(STAGED) and dgr.__file__.startswith(PKG)
    return m


def make_dataset(P: int, width: int, height: int, n_cams: int = 4, seed: int = 0, gt: str = "smooth"):
    """Synthetic `Dataset` (types.py:264-284): pinhole cameras on a small arc (base camera of SURVEY 8(d), yawed by k*5 degrees
    and shifted sideways so that `get_cameras_extent` is non-zero), the SURVEY 8(d) point cloud as SfM points."""
    import wg_scenes as S
    from wildgaussians.types import camera_model_to_int, new_cameras
    cloud = S.make_cloud(P, width, height, sh_degree=None, seed=seed)
    fx = 0.5 * width / math.tan(math.radians(60.0) * 0.5)
    poses = []
    for k in range(n_cams):
        a = math.radians(5.0 * (k - (n_cams - 1) / 2))
        c2w = np.array([[math.cos(a), 0, math.sin(a), 0.3 * (k - (n_cams - 1) / 2)],
                        [0, 1, 0, 0],
                        [-math.sin(a), 0, math.cos(a), 0]], dtype=np.float32)
        poses.append(c2w)
    cams = new_cameras(
        poses=np.stack(poses), intrinsics=np.tile(np.array([[fx, fx, width / 2.0, height / 2.0]], dtype=np.float32), (n_cams, 1)),
        camera_models=np.full((n_cams,), camera_model_to_int("pinhole"), dtype=np.int32),
        distortion_parameters=np.zeros((n_cams, 0), dtype=np.float32),
        image_sizes=np.tile(np.array([[width, height]], dtype=np.int32), (n_cams, 1)), nears_fars=None)
    rng = np.random.default_rng(seed + 7)
    yy, xx = np.mgrid[0:height, 0:width]
    images = []
    for k in range(n_cams):
        if gt == "smooth":   # a smooth target every view agrees on, so that the loss falls within a few steps
            img = np.stack([0.5 + 0.3 * np.sin(xx / width * 3.0 + k * 0.1), 0.5 + 0.3 * np.cos(yy / height * 2.0), 0.4 + 0.0 * xx], -1)
        else:
            img = rng.uniform(0, 1, size=(height, width, 3))
        images.append((np.clip(img, 0, 1) * 255).astype(np.uint8))
    return dict(cameras=cams, image_paths=[f"{k}.png" for k in range(n_cams)], image_paths_root="", mask_paths=None,
                mask_paths_root=None, metadata={}, masks=None, images=images,
                points3D_xyz=cloud["means3D"].astype(np.float32),
                points3D_rgb=(cloud["colors_precomp"] * 255).astype(np.uint8), images_points3D_indices=None), cloud


def make_method(P: int, width: int, height: int, n_cams: int = 4, seed: int = 0, cloud_shapes: str = "knn", overrides=None,
                gt: str = "smooth"):
    """Construct the reference's `WildGaussians` (method.py:1638-1691) for training on the synthetic dataset.
    cloud_shapes = "knn": scales / rotations / opacities exactly as `initialize_from_points3D` sets them (distCUDA2 of this repo);
    "bench": afterwards overwrite those three parameters' DATA with the SURVEY 8(d) cloud's (through the inverse activations), so
    that the rasterizer workload equals the one the restated step bench used (scripts/bench_wildgaussians_step.py)."""
    import torch
    m = import_method()
    ds, cloud = make_dataset(P, width, height, n_cams, seed, gt)
    ov = {"config": "default.yml", "uncertainty_mode": "disabled", "num_sky_gaussians": 0}
    ov.update(overrides or {})
    wg = m.WildGaussians(train_dataset=ds, config_overrides=ov)
    if cloud_shapes == "bench":
        dev = wg.model.xyz.device
        with torch.no_grad():
            n = cloud["scales"].shape[0]
            wg.model.scales.data[:n].copy_(torch.log(torch.from_numpy(cloud["scales"]).to(dev)))
            wg.model.rotations.data[:n].copy_(torch.from_numpy(cloud["rotations"]).to(dev))
            wg.model.opacities.data[:n].copy_(torch.special.logit(torch.from_numpy(cloud["opacities"]).to(dev)))
    return m, wg


class RasterizerTap:
    """Records every `GaussianRasterizer.forward` call (inputs, settings, outputs) through torch's global module hooks --
    the caller and the operator stay untouched."""

    def __init__(self, method_module, grads: bool = False):
        self.cls = method_module.GaussianRasterizer
        self.calls = []
        self._h = None
        self.grads = grads   # also record dL/d(image) of every call as it arrives in the backward pass (call["grad_out"])

    def __enter__(self):
        import torch

        def hook(mod, args, kwargs, out):
            if isinstance(mod, self.cls):
                call = dict(kwargs={k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in kwargs.items()},
                            settings=mod.raster_settings, out=tuple(o.detach().clone() if torch.is_tensor(o) else o for o in out))
                self.calls.append(call)
                if self.grads and torch.is_tensor(out[0]) and out[0].requires_grad:
                    out[0].register_hook(lambda g, call=call: call.__setitem__("grad_out", g.detach().clone()))
        self._h = torch.nn.modules.module.register_module_forward_hook(hook, with_kwargs=True)
        return self

    def __exit__(self, *a):
        self._h.remove()


def reference_build_forward(call, variant="nofma"):
    """The same rasterizer call through oracle/_ref (the reference's own CUDA sources built for gfx950)."""
    import torch
    from oracle.ref_hip import ref_hip
    kw, rs = call["kwargs"], call["settings"]
    cloud = dict(means3D=kw["means3D"], opacities=kw["opacities"], scales=kw["scales"], rotations=kw["rotations"],
                 colors_precomp=kw["colors_precomp"])
    cloud = {k: v.float().cpu().numpy() for k, v in cloud.items()}
    cam = dict(width=rs.image_width, height=rs.image_height, tanfovx=rs.tanfovx, tanfovy=rs.tanfovy,
               viewmatrix=rs.viewmatrix.float().cpu().numpy(), projmatrix=rs.projmatrix.float().cpu().numpy(),
               campos=rs.campos.float().cpu().numpy())
    s = ref_hip.Session(cloud, cam, sh_degree=None, kernel_size=rs.kernel_size, bg=rs.bg.cpu().numpy(),
                        scale_modifier=rs.scale_modifier, subpixel_offset=rs.subpixel_offset.cpu().numpy(), variant=variant)
    R = s.forward()
    torch.cuda.synchronize()
    return dict(num_rendered=R, color=s.color, radii=s.radii, accumulation=(1.0 - s.final_T).reshape(rs.image_height, rs.image_width))
