#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REFERENCE's own Python code.

Run in the build container only (needs /root/reference, which does not exist on the GPU box):

    python tests/golden/make_golden.py

The reference has no tests and no CPU rasterizer, so the only pieces of the
hot path that exist twice in the reference -- once in CUDA, once in importable Python -- are the
"in-repo duplicate formulas" of SURVEY.md section 4:

  * eval_sh                       wildgaussians/method.py:493-548  == computeColorFromSH  forward.cu:20-71
  * build_rotation                wildgaussians/method.py:619-640  == quaternion->R        forward.cu:145-149
  * getWorld2View2 / getProjectionMatrixFromOpenCV / focal2fov
                                  wildgaussians/method.py:587-616  == the matrix conventions the kernels index
  * camera_project                wildgaussians/method.py:86-110   == projection to pixels (ndc2Pix, auxiliary.h:41)
  * compute_3D_filter's near test wildgaussians/method.py:1152-1166 == in_frustum z>0.2    auxiliary.h:154

This script imports wildgaussians.method (with inert stand-ins for the packages that are not
installed here: omegaconf, plyfile, and the two CUDA extensions, none of which are exercised),
evaluates those functions on seeded inputs and stores inputs + outputs as small .npz fixtures.
tests/test_oracle.py checks the CPU oracle against them.  The splatting/compositing arithmetic itself
has no second statement anywhere in the reference; that part is pinned by running the reference's CUDA sources themselves
(hipcc build) on a GPU box: tests/golden/make_golden_ref_hip.py.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def _install_shims():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    mod("omegaconf", OmegaConf=type("OmegaConf", (), {}))
    mod("plyfile", PlyData=type("PlyData", (), {}), PlyElement=type("PlyElement", (), {}))
    sk = mod("simple_knn")
    sk._C = mod("simple_knn._C", distCUDA2=lambda *a, **k: (_ for _ in ()).throw(RuntimeError("shim")))
    mod("diff_gaussian_rasterization", GaussianRasterizationSettings=object, GaussianRasterizer=object)


def main():
    _install_shims()
    sys.path.insert(0, REF)
    from wildgaussians import method as ref  # noqa: E402

    rng = np.random.default_rng(1234)

    # ---- 1. SH evaluation -------------------------------------------------------------------
    P = 257
    means = rng.normal(size=(P, 3)).astype(np.float32) * 3.0
    campos = np.array([0.3, -0.2, 0.1], dtype=np.float32)
    sh = rng.normal(0, 0.4, size=(P, 16, 3)).astype(np.float32)
    dirs = torch.nn.functional.normalize(torch.from_numpy(means) - torch.from_numpy(campos)[None], dim=1)  # method.py:1557
    out = {"means": means, "campos": campos, "sh": sh}
    for deg in range(4):
        shs_view = torch.from_numpy(sh).transpose(1, 2).contiguous()  # [P,3,16]  method.py:1563
        rgb = torch.clamp_min(ref.eval_sh(deg, shs_view, dirs) + 0.5, 0.0)  # method.py:1564-1565
        out[f"rgb_deg{deg}"] = rgb.numpy()
    np.savez(os.path.join(HERE, "sh_eval.npz"), **out)

    # ---- 2. quaternion -> rotation, Sigma = R S S^T R^T ---------------------------------------
    q = rng.normal(size=(64, 4)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    Rm = ref.build_rotation(torch.from_numpy(q), "cpu").numpy()
    s = np.exp(rng.normal(-2, 0.5, size=(64, 3))).astype(np.float32)
    L = Rm.astype(np.float64) * s[:, None, :].astype(np.float64)  # R @ diag(s)
    Sigma = L @ np.transpose(L, (0, 2, 1))
    cov6 = np.stack([Sigma[:, 0, 0], Sigma[:, 0, 1], Sigma[:, 0, 2], Sigma[:, 1, 1], Sigma[:, 1, 2], Sigma[:, 2, 2]], 1)
    np.savez(os.path.join(HERE, "rotation.npz"), q=q, R=Rm, scales=s, cov3D=cov6)

    # ---- 3. camera matrices + projection + near cull ----------------------------------------
    cams = []
    for k, (w, h) in enumerate([(256, 256), (640, 480), (1920, 1080)]):
        a = 0.2 * k
        c2w = np.array([[np.cos(a), 0, np.sin(a), 0.1 * k], [0, 1, 0, -0.05 * k], [-np.sin(a), 0, np.cos(a), 0.2 * k],
                        [0, 0, 0, 1]], dtype=np.float64)
        fx = 0.5 * w / np.tan(np.radians(60.0) * 0.5) * (1.0 + 0.05 * k)
        fy = fx * (1.0 - 0.02 * k)
        cx, cy = w / 2.0 + 3.0 * k, h / 2.0 - 2.0 * k
        # method.py:1502-1525
        pose = np.linalg.inv(c2w)
        R = np.transpose(pose[:3, :3])
        T = pose[:3, 3]
        wv = torch.tensor(ref.getWorld2View2(R, T, np.array([0.0, 0.0, 0.0], dtype=np.float32), 1.0)).transpose(0, 1)
        pm = ref.getProjectionMatrixFromOpenCV(w, h, fx, fy, cx, cy, 0.01, 100.0).transpose(0, 1)
        full = (wv.unsqueeze(0).bmm(pm.unsqueeze(0))).squeeze(0)
        center = wv.inverse()[3, :3]
        tanx = np.tan(ref.focal2fov(float(fx), float(w)) * 0.5)
        tany = np.tan(ref.focal2fov(float(fy), float(h)) * 0.5)
        pts = (rng.normal(size=(200, 3)) * np.array([2.0, 2.0, 3.0]) + np.array([0, 0, 4.0])).astype(np.float32)

        class _Cam:  # the three attributes camera_project reads
            poses = torch.from_numpy(c2w[:3, :4].astype(np.float32))
            intrinsics = torch.tensor([fx, fy, cx, cy], dtype=torch.float32)
        # camera_project expects world->camera as (rotation * uvw).sum(-2) with poses = c2w
        uv = ref.camera_project(_Cam, torch.from_numpy(pts)).numpy()
        # compute_3D_filter's camera-space transform and near test (method.py:1152-1166)
        Rt = torch.tensor(R, dtype=torch.float32)
        Tt = torch.tensor(T, dtype=torch.float32)
        xyz_cam = torch.from_numpy(pts) @ Rt + Tt[None, :]
        cams.append(dict(w=w, h=h, fx=fx, fy=fy, cx=cx, cy=cy, c2w=c2w, viewmatrix=wv.numpy(), projmatrix=full.numpy(),
                         campos=center.numpy(), tanfovx=tanx, tanfovy=tany, pts=pts, uv=uv,
                         xyz_cam=xyz_cam.numpy(), valid_depth=(xyz_cam[:, 2] > 0.2).numpy()))
    flat = {}
    for i, c in enumerate(cams):
        for k, v in c.items():
            flat[f"cam{i}_{k}"] = np.asarray(v)
    np.savez(os.path.join(HERE, "camera.npz"), **flat)
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
