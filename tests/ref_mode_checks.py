"""Checkers of the operator's call modes beyond the reference's surface -- two colour sets in one call (`colors_precomp2=`), two tones of
one SH block (`sh_second=`), one tone (`sh_mul=` / `sh_offset=`), raw-parameter mode (`filter_3D=`) -- against the REFERENCE'S OWN
KERNELS (oracle/_ref, the -ffp-contract=off build), one reference run per image the mode produces.  TEST INFRASTRUCTURE: used by
tests/test_reference_modes.py (inside the driver-run `-m gpu` suite) and by the long sweeps under tests/tools/.

What each mode replaces in the caller, and therefore what it is compared with:
  * two colours / two tones: the two GaussianRasterizer calls of wildgaussians/method.py:1573-1611 -> each image against the reference's
    run with that colour set; each colour input's gradient against that run's; the geometry gradients (means3D, means2D with its
    abs-gradient column, opacities, scales, rotations / cov3D) against the SUM of the two runs' -- what autograd adds over two calls;
  * tones: `(features.clamp_max(1) * mul + offset).clamp_max(1)` of method.py:890-900, 1590-1595 -> the reference is fed the toned
    coefficients built on the host (float32 min, one multiply, one add, min), its dL/dsh chained through that tone on the host;
  * raw parameters: get_gaussians() of method.py:1060-1086 -> the reference is fed the activated parameters, its gradients chained
    through the activations' backward pass.

Bars: radii bit-exact; n_contrib-dependent outputs (image, accumulation) with NO pixel over 1e-4 and the image within 2e-6 (decision-exact
compositing: what is left are the colour sums' fused multiply-adds); accumulation bit-exact; every gradient within 1e-4 of its array's
largest magnitude (north_star: 1e-3; observed <= 1e-5)."""
from __future__ import annotations

import contextlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, os.path.join(ROOT, "wild-gaussians_amd"), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

import wg_scenes as S  # noqa: E402
from wg_testlib import make_settings, rel_err, sweep_case, to_dev  # noqa: E402

IMG_ATOL = 2e-6
GRAD_RTOL = 1e-4


def need_ref(variant: str = "nofma"):
    """The reference build must be there wherever there is a GPU to run it on: a GPU box without oracle/_ref FAILS (a green suite must
    mean the reference was consulted); without a GPU these tests are not selected at all."""
    import pytest
    from oracle.ref_hip import ref_hip
    if not ref_hip.available(variant):
        if torch.cuda.is_available():
            pytest.fail(f"oracle/_ref/{ref_hip.VARIANTS[variant]} is missing on a box with a HIP device: build it where /root/reference exists "
                        "(python __graft_entry__.py) -- it travels with the tree")
        pytest.skip("no HIP device")
    return ref_hip


def to_cov3d(cloud, scale_modifier):
    """Replace the scale / rotation pair by the covariances the kernels would build from them (forward.cu:127-159), in float64."""
    s, q = cloud.pop("scales").astype(np.float64), cloud.pop("rotations").astype(np.float64)
    r, x, y, z = q.T
    Rm = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                   2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                   2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], axis=1).reshape(-1, 3, 3)
    M = Rm * (s * scale_modifier)[:, None, :]
    Sg = M @ M.transpose(0, 2, 1)
    cloud["cov3D_precomp"] = np.stack([Sg[:, 0, 0], Sg[:, 0, 1], Sg[:, 0, 2], Sg[:, 1, 1], Sg[:, 1, 2], Sg[:, 2, 2]], 1).astype(np.float32)
    return cloud


def mode_case(i: int, colours: str, pmod=None, geometry: str = "as_is", sh_degree=None):
    """Case i of the argument sweep of tests/test_parity_gpu.py (_sweep_case: odd frame sizes, fields of view, rotated cameras, mip-filter
    sizes, backgrounds, sub-pixel offsets, scale modifiers), reshaped for a call mode:
    colours "precomp" | "sh" (SH degree `sh_degree`, default the case's own or i % 4); pmod: the cloud is cut to P = pmod (mod 4);
    geometry "pair" | "cov3D" | "as_is"."""
    cloud, cam, deg, kw, W, H = sweep_case(i)
    P = cloud["means3D"].shape[0]
    rng = np.random.default_rng(40_000 + i)
    if colours == "precomp" and "shs" in cloud:
        cloud.pop("shs")
        cloud["colors_precomp"] = rng.uniform(0, 1, size=(P, 3)).astype(np.float32)
        deg = 0
    elif colours == "sh":
        want = sh_degree if sh_degree is not None else (deg if "shs" in cloud else i % 4)
        if "shs" not in cloud or want != deg:
            cloud.pop("colors_precomp", None)
            M = (want + 1) ** 2
            sh = rng.normal(0.0, 0.1, size=(P, M, 3))
            sh[:, 0, :] = rng.normal(0.0, 0.5, size=(P, 3))
            cloud["shs"] = sh.astype(np.float32)
            deg = want
    if geometry == "pair" and "cov3D_precomp" in cloud:   # the sweep's every sixth case carries covariances: take the next one's recipe instead
        raise ValueError(f"case {i} carries covariances; pick another index for a scale / rotation pair")
    if geometry == "cov3D" and "cov3D_precomp" not in cloud:
        to_cov3d(cloud, kw["scale_modifier"])
    if pmod is not None:
        P2 = P - ((P - pmod) % 4)
        cloud = {k: np.ascontiguousarray(v[:P2]) for k, v in cloud.items()}
    return cloud, cam, deg, kw, W, H


PATHS = ("default", "lazy_tiny_fronts", "staged_multi_pass", "global_sort", "band_lists")


@contextlib.contextmanager
def binning_path(path: str):
    """The binning paths the defaults would not take at sweep sizes (tests/test_parity_gpu.py: test_argument_sweep_on_the_alternative_...)."""
    from diff_gaussian_rasterization import _C
    try:
        if path == "lazy_tiny_fronts":
            for k, v in dict(lazy_min_len=256, lazy_target=40, lazy_cap=64).items():
                _C.set_option(k, v)
        elif path == "staged_multi_pass":
            _C.set_option("staged_scatter", 1)
            _C.set_option("staged_scatter_cap", 7)
        elif path == "band_lists":
            _C.set_option("band_list_min_p", 1)
            _C.set_option("staged_scatter", 0)
        elif path == "global_sort":
            _C.set_option("force_global_sort", 1)
        elif path != "default":
            raise ValueError(path)
        yield
    finally:
        for k, v in dict(lazy_sort=1, lazy_min_len=1024, lazy_target=820, lazy_cap=2048, staged_scatter=-1, staged_scatter_cap=0, force_global_sort=0,
                         band_list_min_p=2000000).items():
            _C.set_option(k, v)


def _image_report(img, ref, what):
    a = img.detach().cpu().numpy()
    err = np.abs(a.astype(np.float64) - ref["color"]).max(axis=0)
    return {what + "_pixels_over_1e-4": int((err > 1e-4).sum()), what + "_max": float(err.max())}


def _geometry_grads(t, m2d, r1, r2):
    g = {}
    for k, kk in (("means3D", "means3D"), ("opacities", "opacities"), ("scales", "scales"), ("rotations", "rotations"), ("cov3D_precomp", "cov3Ds_precomp")):
        if k in t:
            g[k] = rel_err(t[k].grad.cpu().numpy().reshape(r1["grads"][kk].shape), r1["grads"][kk] + r2["grads"][kk])
    got, want = m2d.grad.cpu().numpy(), r1["grads"]["means2D"] + r2["grads"]["means2D"]
    g["means2D_xy"] = rel_err(got[:, :2], want[:, :2])
    g["means2D_abs_column"] = rel_err(got[:, 2], want[:, 2])   # GOF abs-gradient (backward.cu:593-595): |g1| + |g2| over two calls, not |g1 + g2|
    return g


def _verdict(rep, radii, ref_radii, acc, ref):
    rep["radii_equal"] = bool(np.array_equal(radii.cpu().numpy(), ref_radii))
    rep["accumulation_bits_equal"] = bool(np.array_equal(acc.detach().cpu().numpy(), (np.float32(1.0) - ref["final_T"].astype(np.float32))))
    rep["worst_grad"] = max(rep["grads"], key=rep["grads"].get)
    rep["ok"] = (rep["radii_equal"] and rep["accumulation_bits_equal"] and all(v == 0 for k, v in rep.items() if k.endswith("_pixels_over_1e-4"))
                 and all(v <= IMG_ATOL for k, v in rep.items() if k.endswith("_max")) and rep["grads"][rep["worst_grad"]] <= GRAD_RTOL)
    return rep


def check_two_colour(case, ref_hip, seed=0):
    """`colors_precomp2=` (wg_second_image): one call, two images, beside two runs of the reference."""
    from diff_gaussian_rasterization import GaussianRasterizer
    cloud, cam, deg, kw, W, H = case
    assert "colors_precomp" in cloud
    P = cloud["means3D"].shape[0]
    c2 = np.random.default_rng(7000 + seed).uniform(0, 1, size=(P, 3)).astype(np.float32)
    cot1, cot2 = S.make_cotangent(W, H, seed=3000 + seed), S.make_cotangent(W, H, seed=5000 + seed)
    r1 = ref_hip.run_scene(cloud, cam, sh_degree=0, cotangent=cot1, variant="nofma", **kw)
    r2 = ref_hip.run_scene(dict(cloud, colors_precomp=c2), cam, sh_degree=0, cotangent=cot2, variant="nofma", **kw)
    rs = make_settings(cam, 0, kw["kernel_size"], kw["bg"], kw["subpixel_offset"], kw["scale_modifier"])
    t = {k: to_dev(v).requires_grad_(True) for k, v in cloud.items()}
    tc2 = to_dev(c2).requires_grad_(True)
    m2d = torch.zeros((P, 3), device="cuda", requires_grad=True)
    geo = {k: t[k] for k in ("scales", "rotations", "cov3D_precomp") if k in t}
    img1, radii, acc, img2 = GaussianRasterizer(rs)(means3D=t["means3D"], means2D=m2d, opacities=t["opacities"], colors_precomp=t["colors_precomp"],
                                                   colors_precomp2=tc2, **geo)
    ((img1 * to_dev(cot1)).sum() + (img2 * to_dev(cot2)).sum()).backward()
    rep = {"mode": "two_colour", "P": P, **_image_report(img1, r1, "img1"), **_image_report(img2, r2, "img2")}
    rep["grads"] = dict(colors1=rel_err(t["colors_precomp"].grad.cpu().numpy(), r1["grads"]["colors_precomp"]),
                        colors2=rel_err(tc2.grad.cpu().numpy(), r2["grads"]["colors_precomp"]), **_geometry_grads(t, m2d, r1, r2))
    return _verdict(rep, radii, r1["radii"], acc, r1)


def tone_host(sh, mul, off, pre, post):
    """(toned coefficients, chain); chain(dL/dtoned) -> (dL/dsh, dL/dmul, dL/doffset).  The tone as include/wg_rasterizer.h defines it:
    x = min(sh, pre); t = x * mul + [k == 0] offset (separate multiply and add); used = min(t, post)."""
    f = np.float32
    x = np.minimum(sh, f(pre))
    t = (x * mul[:, None, :]).astype(f)
    t[:, 0, :] = (t[:, 0, :] + off).astype(f)
    used = np.minimum(t, f(post))

    def chain(g):
        g = np.where(t <= f(post), g, f(0)).astype(np.float64)
        dmul = (g * x).sum(axis=1)
        doff = g[:, 0, :]
        dsh = np.where(sh <= f(pre), g * mul[:, None, :], 0.0)
        return dsh, dmul, doff
    return used.astype(f), chain


def check_two_tone(case, ref_hip, seed=0, second_plain=False):
    """`sh_second=True` (wg_forward_args::sh_second): both tones of one SH block in one call, beside two runs of the reference on the toned
    coefficients.  second_plain: WildGaussians' own shape -- the second set is the clamped coefficients alone."""
    from diff_gaussian_rasterization import GaussianRasterizer
    cloud, cam, deg, kw, W, H = case
    assert "shs" in cloud
    P, M = cloud["shs"].shape[:2]
    rng = np.random.default_rng(11000 + seed)
    sh = (cloud["shs"] * np.float32(rng.choice([1.0, 3.0]))).astype(np.float32)
    mul1, off1 = rng.uniform(0.5, 1.5, size=(P, 3)).astype(np.float32), rng.normal(0, 0.3, size=(P, 3)).astype(np.float32)
    mul2, off2 = rng.uniform(0.8, 1.2, size=(P, 3)).astype(np.float32), rng.normal(0, 0.1, size=(P, 3)).astype(np.float32)
    pre1, post1, pre2, post2 = 1.0, 1.0, float(rng.choice([1.0, 0.3])), float(rng.choice([np.inf, 0.5]))
    if second_plain:
        mul2, off2, post2 = np.ones_like(mul2), np.zeros_like(off2), np.inf
    sh1, chain1 = tone_host(sh, mul1, off1, pre1, post1)
    sh2, chain2 = tone_host(sh, mul2, off2, pre2, post2)
    cot1, cot2 = S.make_cotangent(W, H, seed=3000 + seed), S.make_cotangent(W, H, seed=5000 + seed)
    r1 = ref_hip.run_scene(dict(cloud, shs=sh1), cam, sh_degree=deg, cotangent=cot1, variant="nofma", **kw)
    r2 = ref_hip.run_scene(dict(cloud, shs=sh2), cam, sh_degree=deg, cotangent=cot2, variant="nofma", **kw)
    rs = make_settings(cam, deg, kw["kernel_size"], kw["bg"], kw["subpixel_offset"], kw["scale_modifier"])
    t = {k: to_dev(v).requires_grad_(True) for k, v in dict(cloud, shs=sh).items()}
    tn = dict(mul1=to_dev(mul1).requires_grad_(True), off1=to_dev(off1).requires_grad_(True))
    if not second_plain:
        tn.update(mul2=to_dev(mul2).requires_grad_(True), off2=to_dev(off2).requires_grad_(True))
    m2d = torch.zeros((P, 3), device="cuda", requires_grad=True)
    geo = {k: t[k] for k in ("scales", "rotations", "cov3D_precomp") if k in t}
    second = dict(sh_pre_clamp_max2=pre2) if second_plain else dict(sh_mul2=tn["mul2"], sh_offset2=tn["off2"], sh_pre_clamp_max2=pre2,
                                                                    sh_post_clamp_max2=None if post2 == np.inf else post2)
    img1, radii, acc, img2 = GaussianRasterizer(rs)(means3D=t["means3D"], means2D=m2d, opacities=t["opacities"], shs=t["shs"], **geo,
                                                   sh_mul=tn["mul1"], sh_offset=tn["off1"], sh_pre_clamp_max=pre1, sh_post_clamp_max=post1,
                                                   sh_second=True, **second)
    ((img1 * to_dev(cot1)).sum() + (img2 * to_dev(cot2)).sum()).backward()
    rep = {"mode": "two_tone", "P": P, "deg": deg, "M": M, **_image_report(img1, r1, "img1"), **_image_report(img2, r2, "img2")}
    d1, dm1, do1 = chain1(r1["grads"]["sh"])
    d2, dm2, do2 = chain2(r2["grads"]["sh"])
    g = dict(sh=rel_err(t["shs"].grad.cpu().numpy(), (d1 + d2).astype(np.float32)), mul1=rel_err(tn["mul1"].grad.cpu().numpy(), dm1.astype(np.float32)),
             off1=rel_err(tn["off1"].grad.cpu().numpy(), do1.astype(np.float32)))
    if not second_plain:
        g.update(mul2=rel_err(tn["mul2"].grad.cpu().numpy(), dm2.astype(np.float32)), off2=rel_err(tn["off2"].grad.cpu().numpy(), do2.astype(np.float32)))
    g.update(_geometry_grads(t, m2d, r1, r2))
    rep["grads"] = g
    return _verdict(rep, radii, r1["radii"], acc, r1)


def check_one_tone(case, ref_hip, seed=0):
    """`sh_mul=` / `sh_offset=` / clamps (wg_sh_tone; wg_sh_tone): one toned image beside the reference on the toned coefficients."""
    from diff_gaussian_rasterization import GaussianRasterizer
    cloud, cam, deg, kw, W, H = case
    P, M = cloud["shs"].shape[:2]
    rng = np.random.default_rng(13000 + seed)
    sh = (cloud["shs"] * np.float32(rng.choice([1.0, 3.0]))).astype(np.float32)
    mul, off = rng.uniform(0.5, 1.5, size=(P, 3)).astype(np.float32), rng.normal(0, 0.3, size=(P, 3)).astype(np.float32)
    pre, post = float(rng.choice([1.0, 0.4, np.inf])), float(rng.choice([1.0, 0.6, np.inf]))
    sht, chain = tone_host(sh, mul, off, pre, post)
    cot = S.make_cotangent(W, H, seed=3000 + seed)
    r = ref_hip.run_scene(dict(cloud, shs=sht), cam, sh_degree=deg, cotangent=cot, variant="nofma", **kw)
    rs = make_settings(cam, deg, kw["kernel_size"], kw["bg"], kw["subpixel_offset"], kw["scale_modifier"])
    t = {k: to_dev(v).requires_grad_(True) for k, v in dict(cloud, shs=sh).items()}
    tm, to = to_dev(mul).requires_grad_(True), to_dev(off).requires_grad_(True)
    m2d = torch.zeros((P, 3), device="cuda", requires_grad=True)
    geo = {k: t[k] for k in ("scales", "rotations", "cov3D_precomp") if k in t}
    img, radii, acc = GaussianRasterizer(rs)(means3D=t["means3D"], means2D=m2d, opacities=t["opacities"], shs=t["shs"], **geo, sh_mul=tm, sh_offset=to,
                                            sh_pre_clamp_max=None if pre == np.inf else pre, sh_post_clamp_max=None if post == np.inf else post)
    (img * to_dev(cot)).sum().backward()
    rep = {"mode": "one_tone", "P": P, "deg": deg, **_image_report(img, r, "img1")}
    dsh, dm, do = chain(r["grads"]["sh"])
    zero = {k: np.zeros_like(v) for k, v in r["grads"].items()}
    rep["grads"] = dict(sh=rel_err(t["shs"].grad.cpu().numpy(), dsh.astype(np.float32)), mul=rel_err(tm.grad.cpu().numpy(), dm.astype(np.float32)),
                        off=rel_err(to.grad.cpu().numpy(), do.astype(np.float32)), **_geometry_grads(t, m2d, r, dict(grads=zero)))
    return _verdict(rep, radii, r["radii"], acc, r)


def check_raw(case, ref_hip, seed=0):
    """`filter_3D=` (wg_raw_gaussians): get_gaussians() inside K1 / K11.  The reference is fed the activated parameters -- from this repo's
    stand-alone activation kernels (wg_fused_gaussians.activate: the same device functions; held to the PyTorch restatement of
    method.py:1060-1086 by tests/test_activations.py) --, its gradients are chained through those kernels' backward pass."""
    import wg_fused_gaussians as FG
    from diff_gaussian_rasterization import GaussianRasterizer
    cloud, cam, deg, kw, W, H = case
    assert "scales" in cloud and "colors_precomp" in cloud
    P = cloud["means3D"].shape[0]
    cot = S.make_cotangent(W, H, seed=3000 + seed)
    gen = torch.Generator().manual_seed(9000 + seed)
    filt = (0.3 * torch.rand(P, 1, generator=gen) * torch.from_numpy(cloud["scales"]).mean(dim=1, keepdim=True)).cuda()
    raw = dict(opacities=torch.special.logit(torch.from_numpy(cloud["opacities"]).clamp(1e-4, 1 - 1e-4)).cuda().requires_grad_(True),
               scales=torch.log(torch.from_numpy(cloud["scales"])).cuda().requires_grad_(True),
               rotations=(torch.from_numpy(cloud["rotations"]) * (0.5 + torch.rand(P, 1, generator=gen))).cuda().requires_grad_(True))
    o, s_, q = FG.activate(raw["opacities"], raw["scales"], raw["rotations"], filt)
    r = ref_hip.run_scene(dict(cloud, opacities=o.detach().cpu().numpy(), scales=s_.detach().cpu().numpy(), rotations=q.detach().cpu().numpy()), cam,
                          sh_degree=0, cotangent=cot, variant="nofma", **kw)
    torch.autograd.backward([o, s_, q], [to_dev(r["grads"]["opacities"]).view_as(o), to_dev(r["grads"]["scales"]), to_dev(r["grads"]["rotations"])])
    want = {k: v.grad.clone() for k, v in raw.items()}
    for v in raw.values():
        v.grad = None
    rs = make_settings(cam, 0, kw["kernel_size"], kw["bg"], kw["subpixel_offset"], kw["scale_modifier"])
    m3 = to_dev(cloud["means3D"]).requires_grad_(True)
    col = to_dev(cloud["colors_precomp"]).requires_grad_(True)
    m2d = torch.zeros((P, 3), device="cuda", requires_grad=True)
    img, radii, acc = GaussianRasterizer(rs)(means3D=m3, means2D=m2d, opacities=raw["opacities"], scales=raw["scales"], rotations=raw["rotations"],
                                            colors_precomp=col, filter_3D=filt)
    (img * to_dev(cot)).sum().backward()
    rep = {"mode": "raw", "P": P, **_image_report(img, r, "img1")}
    g = {"raw_" + k: rel_err(raw[k].grad.cpu().numpy(), want[k].cpu().numpy()) for k in raw}
    g["means3D"] = rel_err(m3.grad.cpu().numpy(), r["grads"]["means3D"])
    g["colors"] = rel_err(col.grad.cpu().numpy(), r["grads"]["colors_precomp"])
    got = m2d.grad.cpu().numpy()
    g["means2D_xy"] = rel_err(got[:, :2], r["grads"]["means2D"][:, :2])
    g["means2D_abs_column"] = rel_err(got[:, 2], r["grads"]["means2D"][:, 2])
    rep["grads"] = g
    return _verdict(rep, radii, r["radii"], acc, r)


# ---- geometry outside the reference's domain (NaN / Inf / zero / negative parameters): tests/tools/nonfinite_inputs.py explores, the suite pins ----
_NAN, _INF = np.float32(np.nan), np.float32(np.inf)
NONFINITE_CATEGORIES = {   # name -> (cloud key, component or None = the whole row, value)
    "means x NaN": ("means3D", 0, _NAN), "means z NaN": ("means3D", 2, _NAN), "means x +Inf": ("means3D", 0, _INF), "means y -Inf": ("means3D", 1, -_INF),
    "means z +Inf": ("means3D", 2, _INF), "means z 1e30": ("means3D", 2, np.float32(1e30)),
    "scale NaN": ("scales", 0, _NAN), "scale +Inf": ("scales", 1, _INF), "scale 0": ("scales", 2, np.float32(0)), "scale -1": ("scales", 0, np.float32(-1)),
    "scale 1e20": ("scales", 0, np.float32(1e20)),
    "rot NaN": ("rotations", 0, _NAN), "rot all zero": ("rotations", None, np.float32(0)), "rot +Inf": ("rotations", 1, _INF),
    "opacity NaN": ("opacities", 0, _NAN), "opacity +Inf": ("opacities", 0, _INF), "opacity -1": ("opacities", 0, np.float32(-1)), "opacity 7": ("opacities", 0, np.float32(7)),
}
# The two categories in which the reference's own output is UNDEFINED: a NaN scale or quaternion component makes the 2-D covariance NaN,
# the radius `(int)ceil(NaN)` = 0 and the tile rectangle one tile (forward.cu:241-250) -- the scan allots that instance a slot of the key
# list, but duplicateWithKeys emits nothing for radii <= 0 (rasterizer_impl.cu:85) and the slot keeps whatever the binning buffer held
# before: the reference composites a stale (tile, Gaussian) pair somewhere (tests/tools/debug_nan.py shows its lists one entry short in the
# Gaussian's own tile and the strays in tile 0).  The product counts the same instances (num_rendered agrees), draws nothing for them, and
# reads no uninitialised memory.  Everything else -- sixteen categories, NaN opacities included (alpha = min(0.99f, NaN) = 0.99f over the
# Gaussian's tiles, forward.cu:364) -- the product does bit for bit as the reference's kernels do.
NONFINITE_REFERENCE_UNDEFINED = ("scale NaN", "rot NaN")
NONFINITE_DEVIATING = NONFINITE_REFERENCE_UNDEFINED   # (the name rounds 5's tools use)


def poison(cloud, name, ids):
    """A copy of the cloud with category `name` applied to the Gaussians `ids`."""
    key, comp, val = NONFINITE_CATEGORIES[name]
    c = {k: v.copy() for k, v in cloud.items()}
    if comp is None:
        c[key][ids] = val
    else:
        c[key][ids, comp] = val
    return c
