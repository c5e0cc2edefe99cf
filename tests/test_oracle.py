"""CPU tests of the oracle itself: golden vectors generated from the reference's own Python code
(tests/golden/make_golden.py), float64 finite differences, and structural invariants.  No GPU."""
import os

import numpy as np
import pytest

import wg_scenes as S

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _identity_cam(w=64, h=64):
    return S.make_camera(w, h)


@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_sh_matches_reference_eval_sh(oracle, deg):
    """computeColorFromSH restatement == reference eval_sh (method.py:493-548) + 0.5, clamp."""
    g = np.load(os.path.join(GOLD, "sh_eval.npz"))
    means, campos, sh = g["means"], g["campos"], g["sh"]
    P = means.shape[0]
    # camera placed so that every point is in front of it and lands in the frame is NOT required:
    # rgb is only written for Gaussians that survive culling, so use a camera looking at the cloud
    # from far away along +z and a huge image so nothing is culled.
    shifted = means + np.array([0, 0, 30.0], dtype=np.float32)
    cam = S.make_camera(512, 512)
    cam["campos"] = (campos + np.array([0, 0, 30.0], dtype=np.float32)).astype(np.float32)
    cloud = dict(means3D=shifted, shs=sh, opacities=np.full((P, 1), 0.5, np.float32),
                 scales=np.full((P, 3), 0.05, np.float32), rotations=np.tile(np.array([[1, 0, 0, 0]], np.float32), (P, 1)))
    o = oracle.run_scene(cloud, cam, sh_degree=deg)
    vis = o["radii"] > 0
    assert vis.sum() > 0.9 * P
    rgb = o["ctx"].get("rgb")
    ref = g[f"rgb_deg{deg}"]
    np.testing.assert_allclose(rgb[vis], ref[vis], rtol=0, atol=2e-6)
    clamped = o["ctx"].get("clamped").astype(bool)
    assert (clamped[vis] == (ref[vis] == 0)).mean() > 0.999


def test_cov3d_matches_reference_build_rotation(oracle):
    """computeCov3D restatement == R S S^T R^T with R from the reference build_rotation (method.py:619-640)."""
    g = np.load(os.path.join(GOLD, "rotation.npz"))
    q, s, cov_ref = g["q"], g["scales"], g["cov3D"]
    P = q.shape[0]
    cam = _identity_cam()
    means = np.tile(np.array([[0, 0, 5.0]], np.float32), (P, 1))
    cloud = dict(means3D=means, colors_precomp=np.full((P, 3), 0.5, np.float32), opacities=np.full((P, 1), 0.5, np.float32),
                 scales=s, rotations=q)
    o = oracle.run_scene(cloud, cam, sh_degree=0)
    cov = o["ctx"].get("cov3D")
    np.testing.assert_allclose(cov, cov_ref, rtol=2e-5, atol=1e-7)


@pytest.mark.parametrize("i", [0, 1, 2])
def test_camera_conventions_match_reference(oracle, i):
    """make_camera == method.py:1502-1525; means2D == camera_project - 0.5; near cull == z > 0.2."""
    g = np.load(os.path.join(GOLD, "camera.npz"))
    c = {k[len(f"cam{i}_"):]: g[k] for k in g.files if k.startswith(f"cam{i}_")}
    w, h = int(c["w"]), int(c["h"])
    cam = dict(width=w, height=h, tanfovx=float(c["tanfovx"]), tanfovy=float(c["tanfovy"]),
               viewmatrix=c["viewmatrix"], projmatrix=c["projmatrix"], campos=c["campos"])
    if i == 0:  # symmetric intrinsics: the host-side helper must reproduce the reference's matrices
        mine = S.make_camera(w, h, c2w=c["c2w"])
        np.testing.assert_allclose(mine["viewmatrix"], c["viewmatrix"], atol=1e-6)
        np.testing.assert_allclose(mine["projmatrix"], c["projmatrix"], atol=1e-5)
        np.testing.assert_allclose(mine["campos"], c["campos"], atol=1e-6)
        assert abs(mine["tanfovx"] - cam["tanfovx"]) < 1e-9
    pts = c["pts"]
    P = pts.shape[0]
    cloud = dict(means3D=pts, colors_precomp=np.full((P, 3), 0.5, np.float32), opacities=np.full((P, 1), 0.5, np.float32),
                 scales=np.full((P, 3), 0.01, np.float32), rotations=np.tile(np.array([[1, 0, 0, 0]], np.float32), (P, 1)))
    o = oracle.run_scene(cloud, cam, sh_degree=0)
    vis = oracle.mark_visible(pts, cam["viewmatrix"], cam["projmatrix"])
    assert (vis == c["valid_depth"]).all()
    drawn = o["radii"] > 0
    assert not (drawn & ~vis).any()
    m2d = o["ctx"].get("means2D")
    np.testing.assert_allclose(m2d[drawn], c["uv"][drawn] - 0.5, atol=2e-2 if w > 1000 else 5e-3)
    depths = o["ctx"].get("depths")
    np.testing.assert_allclose(depths[drawn], c["xyz_cam"][drawn, 2], rtol=1e-5, atol=1e-5)


def test_backward_matches_finite_differences_f64(oracle):
    """The restated hand-written backward is the derivative of the restated forward (away from thresholds)."""
    W, H, P = 64, 48, 40
    cam = S.make_camera(W, H)
    cloud = {k: v.astype(np.float64) for k, v in S.make_cloud(P, W, H, sh_degree=3, seed=3, scale_mult=40.0).items()}
    cloud["opacities"] = np.clip(cloud["opacities"], 0.05, 0.6)
    rng = np.random.default_rng(5)
    cot = rng.normal(size=(3, H, W))
    bg = np.array([0.3, 0.2, 0.7])
    so = rng.uniform(-0.5, 0.5, size=(H, W, 2))

    def loss(c):
        return (oracle.run_scene(c, cam, sh_degree=3, bg=bg, subpixel_offset=so, precision="f64")["color"] * cot).sum()

    g = oracle.run_scene(cloud, cam, sh_degree=3, bg=bg, subpixel_offset=so, cotangent=cot, precision="f64")["grads"]
    for key, gk in {"means3D": "means3D", "opacities": "opacities", "shs": "sh", "scales": "scales", "rotations": "rotations"}.items():
        d = rng.normal(size=cloud[key].shape)
        eps = 1e-6
        cp, cm = dict(cloud), dict(cloud)
        cp[key] = cloud[key] + eps * d
        cm[key] = cloud[key] - eps * d
        fd = (loss(cp) - loss(cm)) / (2 * eps)
        an = (g[gk].reshape(d.shape) * d).sum()
        assert abs(fd - an) <= 1e-4 * abs(fd) + 1e-9, (key, fd, an)


def test_binning_invariants(oracle):
    W, H = 160, 96
    cam = S.make_camera(W, H)
    cloud = S.make_cloud(3000, W, H, sh_degree=1, seed=7, scale_mult=4.0)
    o = oracle.run_scene(cloud, cam, sh_degree=1)
    ctx = o["ctx"]
    keys, pl, ranges = ctx.get("keys"), ctx.get("point_list"), ctx.get("ranges")
    R = o["num_rendered"]
    assert R == int(ctx.get("tiles_touched").sum()) == keys.shape[0]
    assert (np.diff(keys.astype(np.uint64)) >= 0).all()  # sorted by (tile | depth)
    # stable: equal keys keep ascending Gaussian index
    same = keys[1:] == keys[:-1]
    assert (pl[1:][same] > pl[:-1][same]).all()
    tiles = (keys >> np.uint64(32)).astype(np.int64)
    for t in np.unique(tiles):
        a, b = ranges[t]
        assert (tiles[a:b] == t).all() and (a == 0 or tiles[a - 1] != t) and (b == R or tiles[b] != t)
    depth_bits = (keys & np.uint64(0xFFFFFFFF)).astype(np.uint32).view(np.float32)
    np.testing.assert_array_equal(depth_bits, ctx.get("depths")[pl])
    assert oracle.get_higher_msb(8160) == 13 and oracle.get_higher_msb(256) == 9 and oracle.get_higher_msb(32400) == 15
    # the product computes it as 32 - clz(n) (csrc/binning.hip: higher_msb): the same function as the reference's bisection
    rng = np.random.default_rng(0)
    for n in list(range(0, 70000)) + [int(x) for x in rng.integers(0, 2 ** 32, size=20000)] + [2 ** k + d for k in range(1, 32) for d in (-1, 0, 1)]:
        assert oracle.get_higher_msb(n) == max(1, int(n).bit_length()), n


def test_empty_and_culled_inputs(oracle):
    cam = _identity_cam(40, 24)
    bg = np.array([0.1, 0.2, 0.3], np.float32)
    # P == 0: zero image, no background (rasterize_points.cu:83)
    e = np.zeros((0, 3), np.float32)
    R, color, radii, ctx = oracle.rasterize_gaussians(bg, e, e, np.zeros((0, 1), np.float32), e, np.zeros((0, 4), np.float32), 1.0,
                                                      np.zeros((0,), np.float32), cam["viewmatrix"], cam["projmatrix"], cam["tanfovx"],
                                                      cam["tanfovy"], 0.1, np.zeros((24, 40, 2), np.float32), 24, 40,
                                                      np.zeros((0,), np.float32), 0, cam["campos"])
    assert R == 0 and color.shape == (3, 24, 40) and not color.any()
    # everything behind the camera: image == background, accumulation == 0
    P = 50
    cloud = S.make_cloud(P, 40, 24, sh_degree=None, seed=1)
    cloud["means3D"][:, 2] = -np.abs(cloud["means3D"][:, 2])
    o = oracle.run_scene(cloud, cam, bg=bg)
    assert o["num_rendered"] == 0 and not (o["radii"] > 0).any()
    np.testing.assert_allclose(o["color"], np.broadcast_to(bg[:, None, None], (3, 24, 40)))
    assert not o["accumulation"].any()
    with pytest.raises(RuntimeError):
        oracle.rasterize_gaussians(bg, np.zeros((5, 2), np.float32), e, e, e, e, 1.0, e, cam["viewmatrix"], cam["projmatrix"], 1, 1, 0.1,
                                   np.zeros((24, 40, 2), np.float32), 24, 40, e, 0, cam["campos"])


def test_plumbing_config1(oracle):
    """BASELINE.json configs[0]: 10k random Gaussians, 256x256, SH deg 0, forward only."""
    cam = S.make_camera(256, 256)
    cloud = S.make_cloud(10000, 256, 256, sh_degree=0)
    o = oracle.run_scene(cloud, cam, sh_degree=0)
    assert o["color"].shape == (3, 256, 256) and np.isfinite(o["color"]).all()
    assert 0 < o["num_rendered"] and 0.5 < (o["radii"] > 0).mean() < 1.0
    acc = o["accumulation"]
    assert acc.min() >= 0 and acc.max() <= 1.0
