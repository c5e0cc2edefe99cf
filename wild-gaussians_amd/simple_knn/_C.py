"""``simple_knn._C`` -- stands where the reference's pybind11 module stands (submodules/simple-knn/ext.cpp:15-17).

    distCUDA2(points: float32[P,3] on a HIP device) -> float32[P]

mean squared distance of each point to its three nearest neighbours (spatial.cu:15-26), computed by the HIP kernels of
``libwg_rasterizer.so`` through the C-ABI of include/wg_knn.h.  No CPU fallback.
"""
import ctypes as C
import os

import torch

_LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "diff_gaussian_rasterization", "libwg_rasterizer.so")
if not os.path.exists(_LIB_PATH):
    raise ImportError(f"{_LIB_PATH} is missing: build it with `python wild-gaussians_amd/build.py` (no CPU fallback)")
_lib = C.CDLL(_LIB_PATH)
_lib.wg_knn_scratch_size.restype = C.c_size_t
_lib.wg_knn_scratch_size.argtypes = [C.c_int]
_lib.wg_knn_mean_dist2.restype = C.c_int
_lib.wg_knn_mean_dist2.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]


def distCUDA2(points: torch.Tensor) -> torch.Tensor:
    if not points.is_cuda:
        raise RuntimeError("distCUDA2: points must live on a HIP device (no CPU path)")
    if points.dim() != 2 or points.size(1) != 3:
        raise RuntimeError("distCUDA2: points must have dimensions (num_points, 3)")
    P = points.size(0)
    pts = points.detach().to(torch.float32).contiguous()
    means = torch.zeros((P,), dtype=torch.float32, device=points.device)  # spatial.cu:21: torch::full({P}, 0.0)
    if P == 0:
        return means
    nbytes = _lib.wg_knn_scratch_size(P)
    scratch = torch.empty((nbytes,), dtype=torch.uint8, device=points.device)
    with torch.cuda.device(points.device):
        status = _lib.wg_knn_mean_dist2(P, pts.data_ptr(), means.data_ptr(), scratch.data_ptr(), nbytes,
                                        torch.cuda.current_stream(points.device).cuda_stream)
    if status != 0:
        raise RuntimeError(f"wg_knn_mean_dist2 failed with status {status}")
    return means
