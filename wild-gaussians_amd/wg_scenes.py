"""Deterministic synthetic Gaussian clouds + cameras for the parity tests and bench.py.

Recipe: SURVEY.md section 8(d) ("Synthetic inputs").  Camera conventions restate how the
reference's caller builds its matrices (wildgaussians/method.py:1502-1525, 605-616, 581-594):
``viewmatrix`` = W2C transposed, ``projmatrix`` = (P @ W2C) transposed, both row-major float32,
i.e. column-major W2C / P.W2C in memory, which is what the rasterizer kernels index.

numpy only (PCG64 streams are stable across numpy versions), so the same bytes are produced in
the build container and on the GPU box.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np


def opencv_projection(w: int, h: int, fx: float, fy: float, cx: float, cy: float, znear: float, zfar: float) -> np.ndarray:
    """Same matrix as getProjectionMatrixFromOpenCV (wildgaussians/method.py:605-616)."""
    P = np.zeros((4, 4), dtype=np.float32)
    P[0, 0] = 2.0 * fx / w
    P[1, 1] = 2.0 * fy / h
    P[0, 2] = (2.0 * cx - w) / w
    P[1, 2] = (2.0 * cy - h) / h
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def make_camera(width: int, height: int, fov_x_deg: float = 60.0, yaw_deg: float = 0.0,
                c2w: Optional[np.ndarray] = None) -> Dict[str, object]:
    """Pinhole camera at the origin looking down +z (OpenCV axes), optionally yawed about +y."""
    fx = 0.5 * width / math.tan(math.radians(fov_x_deg) * 0.5)
    fy = fx
    cx, cy = width / 2.0, height / 2.0
    if c2w is None:
        a = math.radians(yaw_deg)
        c2w = np.array([[math.cos(a), 0, math.sin(a), 0],
                        [0, 1, 0, 0],
                        [-math.sin(a), 0, math.cos(a), 0],
                        [0, 0, 0, 1]], dtype=np.float64)
    w2c = np.linalg.inv(c2w).astype(np.float32)
    proj = opencv_projection(width, height, fx, fy, cx, cy, 0.01, 100.0)
    viewmatrix = np.ascontiguousarray(w2c.T)                       # method.py:1516
    projmatrix = np.ascontiguousarray((viewmatrix @ proj.T))       # method.py:1517-1518
    campos = np.linalg.inv(viewmatrix.astype(np.float64))[3, :3].astype(np.float32)  # method.py:1519
    fovx = 2 * math.atan(width / (2 * fx))                         # focal2fov, method.py:577
    fovy = 2 * math.atan(height / (2 * fy))
    return dict(width=width, height=height, tanfovx=math.tan(fovx * 0.5), tanfovy=math.tan(fovy * 0.5),
                viewmatrix=viewmatrix.astype(np.float32), projmatrix=projmatrix.astype(np.float32),
                campos=np.ascontiguousarray(campos))


def make_cloud(P: int, width: int, height: int, sh_degree: Optional[int] = 3, seed: int = 0,
               fov_x_deg: float = 60.0, scale_mult: float = 1.0) -> Dict[str, np.ndarray]:
    """z~U[1,10]; x,y = z*tanfov*U[-1.1,1.1]; scales 0.002*z*lognormal;
    unit quaternions; opacity U[0.05,0.95]; SH (dc N(0,.5), rest N(0,.1)) or colours U[0,1]."""
    rng = np.random.default_rng(seed)
    tanx = math.tan(math.radians(fov_x_deg) * 0.5)
    tany = tanx * height / width
    z = rng.uniform(1.0, 10.0, size=P)
    u = rng.uniform(-1.1, 1.1, size=P)
    v = rng.uniform(-1.1, 1.1, size=P)
    means = np.stack([z * tanx * u, z * tany * v, z], axis=1)
    s = 0.002 * z * np.exp(rng.normal(0.0, 0.5, size=P))
    scales = s[:, None] * np.exp(rng.normal(0.0, 0.3, size=(P, 3))) * scale_mult
    q = rng.normal(size=(P, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    opac = rng.uniform(0.05, 0.95, size=(P, 1))
    out = dict(means3D=means, scales=scales, rotations=q, opacities=opac)
    if sh_degree is None:
        out["colors_precomp"] = rng.uniform(0.0, 1.0, size=(P, 3))
    else:
        M = (sh_degree + 1) ** 2
        sh = rng.normal(0.0, 0.1, size=(P, M, 3))
        sh[:, 0, :] = rng.normal(0.0, 0.5, size=(P, 3))
        out["shs"] = sh
    return {k: np.ascontiguousarray(a, dtype=np.float32) for k, a in out.items()}


def make_cotangent(width: int, height: int, seed: int = 1) -> np.ndarray:
    """dL/d(out_color) = N(0,1)[3,H,W] / (3HW)  (SURVEY.md 8(d))."""
    rng = np.random.default_rng(seed)
    return (rng.normal(size=(3, height, width)) / (3.0 * height * width)).astype(np.float32)
