"""The compiled torch binding of the C-ABI (wild-gaussians_amd/csrc/torch_binding.cpp -> diff_gaussian_rasterization/_C_torch*.so; INTEGRATION.md
section 2 as a file that builds): the reference's pybind11 surface (ext.cpp:15-19, rasterize_points.h:18-71) on top of libwg_rasterizer.so.
Held to the ctypes binding, which is the default and carries the opt-ins beyond that surface."""
import glob
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wild-gaussians_amd"))
import wg_scenes as S  # noqa: E402

built = bool(glob.glob(os.path.join(ROOT, "wild-gaussians_amd", "diff_gaussian_rasterization", "_C_torch*.so")))
needs_binding = pytest.mark.skipif(not built, reason="compiled binding not built (python wild-gaussians_amd/build.py --torch-binding)")


@pytest.fixture()
def binding():
    from diff_gaussian_rasterization import _C
    before = _C.binding_name()
    yield _C
    _C.use_binding(before)   # (what the process was started with: WG_BINDING, or the default)


@needs_binding
def test_compiled_binding_has_the_reference_modules_surface_and_errors(binding):
    _C = binding
    from diff_gaussian_rasterization import _C_torch as m
    assert {"rasterize_gaussians", "rasterize_gaussians_backward", "mark_visible", "rasterize_gaussians_ex", "rasterize_gaussians_backward_ex"} <= set(dir(m))
    e = torch.Tensor([])
    args = lambda m3: (torch.zeros(3), m3, e, torch.zeros(5, 1), e, e, 1.0, e, torch.eye(4), torch.eye(4), 1.0, 1.0, 0.1, e, 8, 8, e, 0, torch.zeros(3), False, False)
    with pytest.raises(RuntimeError, match="num_points, 3"):      # rasterize_points.cu:59-61
        m.rasterize_gaussians(*args(torch.zeros(5, 2)))
    with pytest.raises(RuntimeError, match="no CPU path"):
        m.rasterize_gaussians(*args(torch.zeros(5, 3)))
    assert _C.binding_name() == "torch"          # the default when it has been built (round 5)
    assert _C.use_binding("ctypes") == "torch" and _C.binding_name() == "ctypes" and _C.use_binding("torch") == "ctypes" and _C.binding_name() == "torch"
    with pytest.raises(ValueError):
        _C.use_binding("pybind")


@pytest.mark.gpu
@needs_binding
@pytest.mark.parametrize("colors", ["sh", "precomp"])
def test_compiled_binding_gives_the_ctypes_bindings_results(binding, colors):
    """Same library underneath: images, radii, accumulation and the image state bit-identical; gradients to the atomics' run-to-run
    rounding; markVisible equal; through the native functions and through the autograd operator."""
    _C = binding
    from diff_gaussian_rasterization import GaussianRasterizer
    from tests.wg_testlib import make_settings, run_hip, run_hip_native, to_dev
    P, W, H = 120_000, 800, 450
    deg = 2 if colors == "sh" else None
    cloud = S.make_cloud(P, W, H, sh_degree=deg, seed=21, scale_mult=2.0)
    cam = S.make_camera(W, H, yaw_deg=-4.0)
    cot = S.make_cotangent(W, H, seed=8)
    d = deg if deg is not None else 0
    out = {}
    for name in ("ctypes", "torch"):
        _C.use_binding(name)
        assert _C.binding_name() == name
        h = run_hip(cloud, cam, sh_degree=d, cotangent=cot)
        n = run_hip_native(cloud, cam, sh_degree=d)
        vis = _C.mark_visible(to_dev(cloud["means3D"]), to_dev(cam["viewmatrix"]), to_dev(cam["projmatrix"]))
        out[name] = dict(h=h, n_contrib=n["views"]["image"]["n_contrib"].cpu().numpy(), final_T=n["views"]["image"]["final_T"].cpu().numpy(),
                         R=int(n["num_rendered"]), vis=vis.cpu().numpy())
    a, b = out["ctypes"], out["torch"]
    assert a["R"] == b["R"] > 0 and np.array_equal(a["vis"], b["vis"]) and a["vis"].dtype == np.bool_
    for k in ("color", "radii", "accumulation"):
        assert np.array_equal(a["h"][k], b["h"][k]), k
    assert np.array_equal(a["n_contrib"], b["n_contrib"]) and np.array_equal(a["final_T"], b["final_T"])
    assert set(a["h"]["grads"]) == set(b["h"]["grads"])
    for k, g in a["h"]["grads"].items():
        assert float(np.abs(g - b["h"]["grads"][k]).max()) <= 4e-6 * float(np.abs(g).max()) + 1e-30, k
    # no Gaussians: zero image, empty radii, nothing launched (rasterize_points.cu:83)
    _C.use_binding("torch")
    rs = make_settings(cam, 0)
    z = lambda *s: torch.zeros(*s, device="cuda")
    img, radii, acc = GaussianRasterizer(rs)(means3D=z(0, 3), means2D=z(0, 3), opacities=z(0, 1), colors_precomp=z(0, 3), scales=z(0, 3), rotations=z(0, 4))
    assert img.shape == (3, H, W) and not img.any() and radii.numel() == 0 and acc.shape == (H, W)


@pytest.mark.gpu
@needs_binding
@pytest.mark.parametrize("name", ["torch", "ctypes"])
def test_wrong_size_tensors_raise_instead_of_reading_out_of_bounds(binding, name):
    """the compiled binding is the default and its fast path skips the Python-side checks, so it repeats them itself -- a
    filter_3D / colors_precomp2 / raw tuple of the wrong length, or filter_3D combined with colors_precomp2, must raise under BOTH
    bindings (the preprocess kernels would otherwise index P entries of a shorter tensor)."""
    _C = binding
    _C.use_binding(name)
    from diff_gaussian_rasterization import GaussianRasterizer
    from tests.wg_testlib import make_settings, to_dev
    P, W, H = 2000, 160, 96
    cloud = S.make_cloud(P, W, H, sh_degree=None, seed=3)
    rast = GaussianRasterizer(make_settings(S.make_camera(W, H), 0))
    t = {k: to_dev(v) for k, v in cloud.items()}
    base = dict(means3D=t["means3D"], means2D=torch.zeros_like(t["means3D"]), opacities=t["opacities"], scales=t["scales"], rotations=t["rotations"])
    col = t["colors_precomp"]
    with pytest.raises((RuntimeError, Exception), match="filter_3D"):
        rast(**base, colors_precomp=col, filter_3D=torch.ones(P - 7, device="cuda"))
    with pytest.raises((RuntimeError, Exception), match="colors2|colors_precomp2"):
        rast(**base, colors_precomp=col, colors_precomp2=col[: P - 1])
    with pytest.raises((RuntimeError, Exception), match="colors2|colors_precomp2"):
        rast(**base, colors_precomp=col[: P - 1].contiguous(), colors_precomp2=col)
    with pytest.raises((RuntimeError, Exception), match="filter_3D"):
        rast(**base, colors_precomp=col, colors_precomp2=col, filter_3D=torch.ones(P, device="cuda"))
    if name == "torch":   # the reference-shaped arguments too (the reference's own binding checks none of them)
        with pytest.raises(RuntimeError, match="scales"):
            rast(**{**base, "scales": t["scales"][: P - 3].contiguous()}, colors_precomp=col)
        with pytest.raises(RuntimeError, match="opacities"):
            rast(**{**base, "opacities": t["opacities"][: P - 3].contiguous()}, colors_precomp=col)
    # the raw tuple of the backward call
    e = torch.Tensor([])
    rs = rast.raster_settings
    R, color, radii, gb, bb, ib = _C.rasterize_gaussians(rs.bg, t["means3D"], col, t["opacities"], t["scales"], t["rotations"], 1.0, e, rs.viewmatrix, rs.projmatrix,
                                                         rs.tanfovx, rs.tanfovy, rs.kernel_size, rs.subpixel_offset, H, W, e, 0, rs.campos, False, False)[:6]
    with pytest.raises(RuntimeError, match="raw"):
        _C.rasterize_gaussians_backward(rs.bg, t["means3D"], radii, col, t["scales"], t["rotations"], 1.0, e, rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy,
                                        rs.kernel_size, rs.subpixel_offset, torch.zeros(3, H, W, device="cuda"), e, 0, rs.campos, gb, R, bb, ib, False,
                                        raw=(torch.ones(P - 5, device="cuda"), t["opacities"]))
    # and a well-formed call still works afterwards
    img = rast(**base, colors_precomp=col)[0]
    assert torch.isfinite(img).all()


def test_binding_follows_the_environment():
    """WG_BINDING=ctypes|torch selects the binding at import; without it the compiled one is the default when it has been built."""
    from diff_gaussian_rasterization import _C
    want = os.environ.get("WG_BINDING") or ("torch" if built and not os.environ.get("WG_RASTERIZER_LIB") else "ctypes")
    assert _C.binding_name() == want


@pytest.mark.gpu
@needs_binding
def test_cross_section_of_the_suite_passes_under_the_ctypes_binding():
    """The ctypes binding (`WG_BINDING=ctypes`: what serves a variant library, and the fallback where the compiled module has not been
    built) is a supported path, so a cross-section of the GPU suite runs under it INSIDE the driver-run suite:
    the four parity cases forward and backward, the operator surface, the thirty-case argument sweep, the deterministic mode, and every call
    mode beyond the reference's surface (two colour sets, two tones, one tone, raw parameters: twelve cases) against the reference's own
    kernels -- in a subprocess whose import picks the binding from the environment."""
    import re
    import subprocess
    sel = ("binding_follows_the_environment or forward_rgb_parity or backward_gradient_parity or operator_surface_semantics or "
           "argument_sweep_against_the_oracle or deterministic_backward_mode_is_bit_reproducible or "
           "(two_colour_sets_in_one_call and (0 or 5 or 11)) or (two_tones_of_one_sh_block and (1 or 6 or 12)) or "
           "(one_tone_beside and (2 or 7)) or (raw_parameter_mode_beside and 0) or wrong_size_tensors")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu or not gpu", "-p", "no:cacheprovider", "tests/test_parity_gpu.py",
                        "tests/test_reference_modes.py", "tests/test_torch_binding.py", "-k", sel],
                       cwd=ROOT, env=dict(os.environ, WG_BINDING="ctypes"), capture_output=True, text=True, timeout=900)
    tail = r.stdout[-3000:] + r.stderr[-1500:]
    assert r.returncode == 0, tail
    m = re.search(r"(\d+) passed", r.stdout)
    assert m and int(m.group(1)) >= 40, tail
