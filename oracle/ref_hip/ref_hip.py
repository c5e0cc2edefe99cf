"""ctypes binding of oracle/_ref/libref_hip_rasterizer*.so -- the REAL reference rasterizer built for gfx950
(oracle/ref_hip/Makefile, driver.cpp).  TEST INFRASTRUCTURE ONLY: used by tests/golden/make_golden_ref_hip.py (fixtures
that pin the CPU oracle), tests/test_reference_hip_gpu.py and bench.py's baseline leg.  Needs a GPU; inputs are numpy
arrays in the wg_scenes layout, outputs numpy (run_scene) -- or torch tensors left on the device (Session, for timing).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_REF_DIR = os.path.join(os.path.dirname(_HERE), "_ref")
VARIANTS = {"default": "libref_hip_rasterizer.so", "nofma": "libref_hip_rasterizer_nofma.so"}
_LIBS = {}


def lib_path(variant: str = "default") -> str:
    return os.path.join(_REF_DIR, VARIANTS[variant])


def available(variant: str = "default") -> bool:
    return os.path.exists(lib_path(variant))


def build() -> bool:
    """Compile from /root/reference where it exists (the build container); elsewhere keep whatever was built."""
    if not os.path.isdir("/root/reference/submodules/diff-gaussian-rasterization/cuda_rasterizer"):
        return available()
    subprocess.run(["make", "-C", _HERE, "-s"], check=True)
    return True


def _lib(variant):
    if variant not in _LIBS:
        lib = C.CDLL(lib_path(variant))
        p, i, f = C.c_void_p, C.c_int, C.c_float
        lib.refhip_forward.restype = i
        lib.refhip_forward.argtypes = [i, i, i, p, i, i, p, p, p, p, p, f, p, p, p, p, p, f, f, f, p, i, p, p, p, p, i]
        lib.refhip_backward.restype = None
        lib.refhip_backward.argtypes = [i, i, i, p, i, i, p, p, p, p, f, p, p, p, p, p, f, f, f, p, p, p] + [p] * 9 + [i]
        lib.refhip_mark_visible.restype = None
        lib.refhip_mark_visible.argtypes = [i, p, p, p, p]
        lib.refhip_release.restype = None
        _LIBS[variant] = lib
    return _LIBS[variant]


def _dp(t):
    return None if t is None or t.numel() == 0 else C.c_void_p(t.data_ptr())


class Session:
    """One scene resident on the device; forward()/backward() call the reference and leave results on the device."""

    def __init__(self, cloud, cam, sh_degree=3, kernel_size=0.1, bg=None, scale_modifier=1.0, subpixel_offset=None,
                 variant="default", device="cuda"):
        import torch
        self.torch, self.lib = torch, _lib(variant)
        dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(device)
        self.H, self.W = cam["height"], cam["width"]
        self.P = cloud["means3D"].shape[0]
        self.t = {k: dev(v) for k, v in cloud.items()}
        self.M = self.t["shs"].shape[1] if "shs" in self.t else 0
        self.D = sh_degree if sh_degree is not None else 0
        self.bg = dev(np.zeros(3) if bg is None else bg)
        self.so = dev(np.zeros((self.H, self.W, 2)) if subpixel_offset is None else subpixel_offset)
        self.view, self.proj, self.campos = dev(cam["viewmatrix"]), dev(cam["projmatrix"]), dev(cam["campos"])
        self.tanx, self.tany = float(cam["tanfovx"]), float(cam["tanfovy"])
        self.kernel_size, self.scale_modifier = float(kernel_size), float(scale_modifier)
        z = lambda *s, dt=torch.float32: torch.zeros(s, dtype=dt, device=device)
        P, N = self.P, self.H * self.W
        self.color, self.radii = z(3, self.H, self.W), z(P, dt=torch.int32)
        self.final_T, self.n_contrib = z(N), z(N, dt=torch.int32)
        self.g = dict(means2D=z(P, 3), conic=z(P, 2, 2), opacities=z(P, 1), colors_precomp=z(P, 3), means3D=z(P, 3),
                      cov3Ds_precomp=z(P, 6), sh=z(P, max(self.M, 1), 3), scales=z(P, 3), rotations=z(P, 4))
        self.num_rendered = 0

    def forward(self, copy_image_state=True):
        t = self.t
        self.num_rendered = self.lib.refhip_forward(
            self.P, self.D, self.M, _dp(self.bg), self.W, self.H, _dp(t["means3D"]), _dp(t.get("shs")),
            _dp(t.get("colors_precomp")), _dp(t["opacities"]), _dp(t.get("scales")), self.scale_modifier,
            _dp(t.get("rotations")), _dp(t.get("cov3D_precomp")), _dp(self.view), _dp(self.proj), _dp(self.campos),
            self.tanx, self.tany, self.kernel_size, _dp(self.so), 0, _dp(self.color), _dp(self.radii),
            _dp(self.final_T) if copy_image_state else None, _dp(self.n_contrib) if copy_image_state else None, 0)
        if self.num_rendered < 0:
            raise MemoryError("reference build: scratch allocation failed")
        return self.num_rendered

    def backward(self, dL_dpix):
        t, g = self.t, self.g
        self.lib.refhip_backward(
            self.P, self.D, self.M, _dp(self.bg), self.W, self.H, _dp(t["means3D"]), _dp(t.get("shs")),
            _dp(t.get("colors_precomp")), _dp(t.get("scales")), self.scale_modifier, _dp(t.get("rotations")),
            _dp(t.get("cov3D_precomp")), _dp(self.view), _dp(self.proj), _dp(self.campos), self.tanx, self.tany,
            self.kernel_size, _dp(self.so), _dp(self.radii), _dp(dL_dpix), _dp(g["means2D"]), _dp(g["conic"]),
            _dp(g["opacities"]), _dp(g["colors_precomp"]), _dp(g["means3D"]), _dp(g["cov3Ds_precomp"]), _dp(g["sh"]),
            _dp(g["scales"]), _dp(g["rotations"]), 0)

    def mark_visible(self):
        present = self.torch.zeros(self.P, dtype=self.torch.uint8, device=self.color.device)
        self.lib.refhip_mark_visible(self.P, _dp(self.t["means3D"]), _dp(self.view), _dp(self.proj), _dp(present))
        return present.bool()


def run_scene(cloud, cam, sh_degree=3, kernel_size=0.1, bg=None, scale_modifier=1.0, subpixel_offset=None, cotangent=None,
              variant="default", device="cuda"):
    """Same dictionary as oracle.run_scene / wg_testlib.run_hip, from the reference's own kernels."""
    import torch
    s = Session(cloud, cam, sh_degree, kernel_size, bg, scale_modifier, subpixel_offset, variant, device)
    R = s.forward()
    torch.cuda.synchronize()
    out = dict(num_rendered=R, color=s.color.cpu().numpy(), radii=s.radii.cpu().numpy(),
               final_T=s.final_T.cpu().numpy().reshape(s.H, s.W),
               n_contrib=s.n_contrib.cpu().numpy().reshape(s.H, s.W))
    out["accumulation"] = 1.0 - out["final_T"]
    if cotangent is not None:
        s.backward(torch.from_numpy(np.ascontiguousarray(cotangent, dtype=np.float32)).to(device))
        torch.cuda.synchronize()
        g = {k: v.cpu().numpy() for k, v in s.g.items()}
        if s.M == 0:
            g.pop("sh")
        out["grads"] = g
    out["visible"] = s.mark_visible().cpu().numpy()
    return out


def knn_available() -> bool:
    return os.path.exists(os.path.join(_REF_DIR, "libref_hip_knn.so"))


def dist_cuda2(points, device="cuda", variant="default") -> np.ndarray:
    """The reference's simple_knn distCUDA2 (spatial.cu:15-26) itself: float32[P] mean squared distance to the 3 nearest."""
    import torch
    key = "knn" if variant == "default" else "knn_nofma"
    if key not in _LIBS:
        lib = C.CDLL(os.path.join(_REF_DIR, f"libref_hip_{key}.so"))
        lib.refhip_knn.restype = None
        lib.refhip_knn.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
        _LIBS[key] = lib
    pts = torch.from_numpy(np.ascontiguousarray(points, dtype=np.float32)).to(device)
    out = torch.zeros(pts.shape[0], dtype=torch.float32, device=device)
    torch.cuda.synchronize()
    _LIBS[key].refhip_knn(pts.shape[0], _dp(pts), C.c_void_p(out.data_ptr()))
    torch.cuda.synchronize()
    return out.cpu().numpy()
