"""simple_knn._C.distCUDA2 replacement.  CPU: the oracle's restatement of simple_knn.cu equals an
independent brute-force 3-NN, the C-ABI exports wg_knn.h.  GPU: the HIP path equals the oracle bit for bit."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _clouds():
    rng = np.random.default_rng(42)
    out = {}
    for P in (1, 2, 3, 4, 7, 100, 1024, 1025, 5000):
        out[f"normal_{P}"] = (rng.normal(size=(P, 3)) * np.array([3.0, 1.0, 0.2]) + np.array([1.0, 2.0, 3.0])).astype(np.float32)
    dup = rng.normal(size=(300, 3)).astype(np.float32)
    out["duplicates"] = np.concatenate([dup, dup[:100], dup[:50]], 0)            # exact duplicates: neighbours at distance 0
    flat = rng.uniform(-1, 1, size=(2000, 3)).astype(np.float32)
    flat[:, 2] = 0.0                                                               # degenerate axis: (z - 0) / (0 - 0)
    out["planar_z0"] = flat
    out["negative_octant"] = (-np.abs(rng.normal(size=(3000, 3))) - 5.0).astype(np.float32)  # box must still contain the origin
    out["clustered"] = np.concatenate([rng.normal(size=(3000, 3)) * 0.01 + c for c in rng.uniform(-10, 10, size=(4, 3))], 0).astype(np.float32)
    return out


@pytest.mark.parametrize("name", list(_clouds()))
def test_oracle_restatement_equals_bruteforce(oracle, name):
    pts = _clouds()[name]
    a = oracle.dist_cuda2(pts)
    b = oracle.dist_cuda2(pts, bruteforce=True)
    np.testing.assert_array_equal(a, b)
    if pts.shape[0] >= 4:
        assert np.isfinite(a).all() and (a >= 0).all()


def _reference_golden():
    d = np.load(os.path.join(ROOT, "tests", "golden", "ref_hip_knn_golden.npz"))
    return {str(n): (d[f"{n}/points"], d[f"{n}/mean_dist2"], d[f"{n}/mean_dist2_nofma"]) for n in d["names"]}


@pytest.mark.parametrize("name", list(_clouds()))
def test_oracle_matches_the_reference_itself(oracle, name):
    """tests/golden/ref_hip_knn_golden.npz: the reference's simple_knn.cu itself, compiled for gfx950 with hipcc
    (oracle/ref_hip/Makefile) and run on an MI355X by tests/golden/make_golden_ref_hip.py -- once with the compiler's default
    fp contraction, once with -ffp-contract=off (the arithmetic the source spells).  Bit-exact with the latter; the two
    builds of the reference differ from each other by an ulp (<= 2.4e-7 relative) in 5-10 % of the points."""
    pts, ref, ref_nofma = _reference_golden()[name]
    np.testing.assert_array_equal(pts, _clouds()[name])
    a = oracle.dist_cuda2(pts)
    np.testing.assert_array_equal(a.view(np.uint32), ref_nofma.view(np.uint32))
    fin = np.isfinite(ref)
    assert np.array_equal(np.isfinite(a), fin)
    assert (np.abs(a[fin] - ref[fin]) <= 1e-6 * np.abs(ref[fin])).all()


def test_knn_abi_exported():
    lib = C.CDLL(os.path.join(ROOT, "wild-gaussians_amd", "diff_gaussian_rasterization", "libwg_rasterizer.so"))
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "wg_knn.h")).read(), flags=re.S)
    names = set(re.findall(r"\b(wg_[a-z0-9_]+)\s*\(", text))
    assert names == {"wg_knn_scratch_size", "wg_knn_mean_dist2"}
    for n in names:
        assert hasattr(lib, n)
    lib.wg_knn_scratch_size.restype, lib.wg_knn_scratch_size.argtypes = C.c_size_t, [C.c_int]
    assert lib.wg_knn_scratch_size(1_000_000) >= 1_000_000 * (16 + 16)
    lib.wg_knn_mean_dist2.restype = C.c_int
    lib.wg_knn_mean_dist2.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    assert lib.wg_knn_mean_dist2(-1, None, None, None, 0, None) == -1
    assert lib.wg_knn_mean_dist2(10, None, None, None, 0, None) == -1
    assert lib.wg_knn_mean_dist2(0, None, None, None, 0, None) == 0
    from simple_knn._C import distCUDA2
    with pytest.raises(RuntimeError, match="no CPU path"):
        distCUDA2(torch.zeros(5, 3))


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(_clouds()) + ["uniform_200k"])
def test_distcuda2_matches_oracle_bit_exact(oracle, name):
    from simple_knn._C import distCUDA2
    if name == "uniform_200k":
        pts = np.random.default_rng(7).uniform(-4, 4, size=(200_000, 3)).astype(np.float32)
    else:
        pts = _clouds()[name]
    ref = oracle.dist_cuda2(pts)
    out = distCUDA2(torch.from_numpy(pts).cuda())
    assert out.shape == (pts.shape[0],) and out.dtype == torch.float32
    np.testing.assert_array_equal(out.cpu().numpy().view(np.uint32), ref.view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(_clouds()))
def test_distcuda2_matches_the_reference_itself(name):
    from simple_knn._C import distCUDA2
    pts, ref, ref_nofma = _reference_golden()[name]
    out = distCUDA2(torch.from_numpy(pts).cuda()).cpu().numpy()
    np.testing.assert_array_equal(out.view(np.uint32), ref_nofma.view(np.uint32))
    fin = np.isfinite(ref)
    assert (np.abs(out[fin] - ref[fin]) <= 1e-6 * np.abs(ref[fin])).all()


@pytest.mark.gpu
def test_distcuda2_call_pattern_of_method_py():
    """method.py:1001: dist2 = torch.clamp_min(distCUDA2(points.float().cuda()), 0.0000001)."""
    from simple_knn._C import distCUDA2
    pts = torch.from_numpy(np.random.default_rng(1).normal(size=(10000, 3))).float().cuda()
    d = torch.clamp_min(distCUDA2(pts), 0.0000001)
    assert d.shape == (10000,) and torch.isfinite(d).all()
    assert distCUDA2(torch.zeros(0, 3, device="cuda")).numel() == 0
