"""Pins: the CPU oracle (and, on a GPU, the HIP product) against outputs of the REFERENCE ITSELF.

tests/golden/ref_hip_golden.npz holds, for the seven cases of tests/golden/ref_hip_cases.py, what the reference's own CUDA
sources -- compiled for gfx950 with hipcc by oracle/ref_hip/Makefile and run on an MI355X by
tests/golden/make_golden_ref_hip.py -- return: num_rendered, radii, the image, final_T, n_contrib, markVisible and all
nine gradient arrays (profiles/r1_ref_hip_golden_report.json is the report of that run).  Two builds of the reference
were recorded: compiler-default fp contraction ("default", full outputs) and -ffp-contract=off ("nofma", integers only).

Bars: integer outputs bit-exact with BOTH builds; images within 1e-5 (north_star: 1e-4); gradients within 5e-5 relative to
the array's largest magnitude (north_star: 1e-3).  Observed: 1.4e-6 and 6.3e-6, all of it the fp-contraction difference
between the two builds of the reference itself.
"""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden", "ref_hip_golden.npz")
IMG_ATOL = 1e-5
GRAD_RTOL = 5e-5


def _load():
    d = np.load(GOLDEN)
    cases = []
    for name in d["names"]:
        name = str(name)
        pre = name + "/"
        ins = {k[len(pre) + 3:]: d[k] for k in d.files if k.startswith(pre + "in/")}
        w, h, tanx, tany = d[pre + "cam/scalars"]
        cam = dict(width=int(w), height=int(h), tanfovx=float(tanx), tanfovy=float(tany), viewmatrix=d[pre + "cam/viewmatrix"],
                   projmatrix=d[pre + "cam/projmatrix"], campos=d[pre + "cam/campos"])
        kw = json.loads(str(d[pre + "kw"]))
        for k in ("bg",):
            if k in kw:
                kw[k] = np.asarray(kw[k], np.float32)
        cot = ins.pop("cotangent")
        if "subpixel_offset" in ins:
            kw["subpixel_offset"] = ins.pop("subpixel_offset")
        ref = {k[len(pre) + 8:]: d[k] for k in d.files if k.startswith(pre + "default/") and "/grad/" not in k}
        ref["grads"] = {k.split("/grad/")[1]: d[k] for k in d.files if k.startswith(pre + "default/grad/")}
        nofma = {k[len(pre) + 6:]: d[k] for k in d.files if k.startswith(pre + "nofma/")}
        cases.append((name, ins, cam, kw, cot, ref, nofma))
    return cases


CASES = _load()
IDS = [c[0] for c in CASES]


def _rel(a, ref):
    a, ref = np.asarray(a, np.float64), np.asarray(ref, np.float64)
    return float(np.abs(a.reshape(ref.shape) - ref).max() / (np.abs(ref).max() + 1e-12)) if ref.size else 0.0


def test_fixture_covers_every_argument_of_the_operator():
    assert len(CASES) == 7
    kws = [c[3] for c in CASES]
    assert any("subpixel_offset" in k for k in kws) and any(k.get("kernel_size") == 0.0 for k in kws)
    assert any(k.get("scale_modifier", 1.0) != 1.0 for k in kws) and any("bg" in k for k in kws)
    assert any("cov3D_precomp" in c[1] for c in CASES) and any("colors_precomp" in c[1] for c in CASES)
    assert {c[3]["sh_degree"] for c in CASES} == {0, 1, 2, 3}
    assert any(c[5]["final_T"].min() < 1.1e-4 for c in CASES)  # the T < 1e-4 stop is reached somewhere
    assert any((c[5]["radii"] == 0).sum() > 100 for c in CASES)  # culling


# image tolerance against the reference's -ffp-contract=off build with decision-exact compositing: no decision differs, what is left are
# the fused multiply-adds of the colour sums (a few ulp of a pixel's value)
EXACT_IMG_ATOL = 2e-6


@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_oracle_matches_the_reference_itself(case):
    from oracle import oracle
    name, cloud, cam, kw, cot, ref, nofma = case
    o = oracle.run_scene(cloud, cam, cotangent=cot, **kw)
    H, W = cam["height"], cam["width"]
    n_contrib = np.asarray(o["ctx"].get("n_contrib")).reshape(H, W)
    final_T = np.asarray(o["ctx"].get("final_T")).reshape(H, W)
    for r in (ref, nofma):  # integer outputs: bit-exact with both builds of the reference
        assert int(o["num_rendered"]) == int(r["num_rendered"])
        assert np.array_equal(o["radii"], r["radii"])
        assert np.array_equal(n_contrib.astype(np.int64), r["n_contrib"].astype(np.int64))
    assert np.abs(o["color"] - ref["color"]).max() <= IMG_ATOL
    assert np.abs(final_T - ref["final_T"]).max() <= IMG_ATOL
    vis = oracle.mark_visible(cloud["means3D"], cam["viewmatrix"], cam["projmatrix"])
    assert np.array_equal(vis, ref["visible"].astype(bool))
    for k, g in ref["grads"].items():
        assert k in o["grads"], k
        assert _rel(o["grads"][k], g) <= GRAD_RTOL, (k, _rel(o["grads"][k], g))


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_product_matches_the_reference_itself(case):
    import torch
    from tests.wg_testlib import run_hip, run_hip_native, to_dev
    from diff_gaussian_rasterization import _C
    name, cloud, cam, kw, cot, ref, nofma = case
    h = run_hip(cloud, cam, cotangent=cot, **kw)
    H, W = cam["height"], cam["width"]
    for r in (ref, nofma):
        assert np.array_equal(h["radii"], r["radii"])
    assert np.abs(h["color"] - ref["color"]).max() <= IMG_ATOL
    assert np.abs(h["accumulation"].reshape(H, W) - (1.0 - ref["final_T"])).max() <= IMG_ATOL
    for k, g in ref["grads"].items():
        if k in h["grads"]:  # dL/dconic and (with SH) dL/dcolour are internal to the product
            assert _rel(h["grads"][k], g) <= GRAD_RTOL, (k, _rel(h["grads"][k], g))
    assert {"means3D", "means2D", "opacities"} <= set(h["grads"])
    if kw.get("scale_modifier", 1.0) == 1.0:  # run_hip_native takes no modifier
        n = run_hip_native(cloud, cam, sh_degree=kw["sh_degree"], kernel_size=kw.get("kernel_size", 0.1), bg=kw.get("bg"),
                           subpixel_offset=kw.get("subpixel_offset"))
        assert int(n["num_rendered"]) == int(ref["num_rendered"]) == int(nofma["num_rendered"])
        assert np.array_equal(n["views"]["image"]["n_contrib"].cpu().numpy().reshape(H, W).astype(np.int64), ref["n_contrib"].astype(np.int64))
    present = _C.mark_visible(to_dev(cloud["means3D"]), to_dev(cam["viewmatrix"]), to_dev(cam["projmatrix"]))
    assert np.array_equal(present.cpu().numpy().astype(bool), ref["visible"].astype(bool))


@pytest.mark.gpu
@pytest.mark.parametrize("P,W,H,colors,scale_mult,yaw", [(200_000, 800, 448, "sh", 1.0, 0.0), (100_000, 640, 360, "precomp", 6.0, 15.0),
                                                         (300_000, 1000, 600, "sh", 3.0, -7.0)])
def test_product_beside_the_reference_build_at_sizes_the_cpu_oracle_is_slow_for(P, W, H, colors, scale_mult, yaw):
    """The reference's own kernels (oracle/_ref, prebuilt; both builds) and the product on the same GPU, same inputs.  Against the
    no-contraction build nothing may differ; against the compiler-default build only what differs between the reference's own two
    builds (its fused multiply-adds move a radius or a threshold decision by an ulp: a handful per million)."""
    from ref_mode_checks import need_ref
    ref_hip = need_ref("default")
    need_ref("nofma")
    import wg_scenes as S
    from tests.wg_testlib import run_hip
    deg = 3 if colors == "sh" else None
    cloud = S.make_cloud(P, W, H, sh_degree=deg, seed=3, scale_mult=scale_mult)
    cam = S.make_camera(W, H, yaw_deg=yaw)
    cot = S.make_cotangent(W, H, seed=4)
    d = deg if deg is not None else 0
    r = ref_hip.run_scene(cloud, cam, sh_degree=d, cotangent=cot)                      # compiler-default contraction: what a user's build runs
    rn = ref_hip.run_scene(cloud, cam, sh_degree=d, cotangent=cot, variant="nofma")    # the arithmetic the sources spell
    h = run_hip(cloud, cam, sh_degree=d, cotangent=cot)
    # (1) against the no-contraction build: nothing differs (no flip budget)
    assert np.array_equal(h["radii"], rn["radii"])
    errn = np.abs(h["color"].astype(np.float64) - rn["color"]).max(axis=0)
    assert int((errn > 1e-4).sum()) == 0 and errn.max() <= EXACT_IMG_ATOL, (int((errn > 1e-4).sum()), float(errn.max()))
    assert np.array_equal(h["accumulation"].reshape(H, W), (np.float32(1.0) - rn["final_T"].astype(np.float32)))
    # (2) against the default build: the product may differ from it ONLY where the reference's two builds differ from each other -- a
    # fused multiply-add of that build moving a radius or a threshold decision by an ulp (a handful per million)
    err = np.abs(h["color"].astype(np.float64) - r["color"]).max(axis=0)
    between_builds = np.abs(rn["color"].astype(np.float64) - r["color"]).max(axis=0)
    acc_err = np.abs(h["accumulation"].reshape(H, W) - r["accumulation"])
    print({"radii_mismatch_vs_default_build": int((h["radii"] != r["radii"]).sum()), "pixels_over_1e-4_vs_default_build": int((err > 1e-4).sum()),
           "of_which_the_two_reference_builds_differ": int(((err > 1e-4) & (between_builds > 1e-4)).sum()), "max": float(err.max()),
           "grads": {k: _rel(g, r["grads"][k]) for k, g in h["grads"].items()}})
    assert np.array_equal(h["radii"] != r["radii"], rn["radii"] != r["radii"])
    assert not ((err > 1e-4) & ~(between_builds > 1e-4 - 2 * EXACT_IMG_ATOL)).any()
    assert np.quantile(err, 0.9999) <= 1e-5
    assert np.quantile(acc_err, 0.9999) <= 1e-5
    for k, g in h["grads"].items():
        assert _rel(g, rn["grads"][k]) <= 1e-3, (k, _rel(g, rn["grads"][k]))
        assert _rel(g, r["grads"][k]) <= 1e-3, (k, _rel(g, r["grads"][k]))


@pytest.mark.gpu
@pytest.mark.parametrize("P,W,H,colors,scale_mult,label", [
    (3_000_000, 1600, 1200, "precomp", 1.0, "config 3 size: 3 M Gaussians, 1600x1200, precomputed colours (forward + backward)"),
    (2_000_000, 3840, 2160, "sh", 1.5, "config-5-shaped: 4K, 32 400 tiles -> LDS binning, lazy front sort"),
    (2_000_000, 4096, 2400, "sh", 1.5, "38 400 tiles > BIN_MAX_TILES -> the global-sort binning path"),
    (10_000_000, 3840, 2160, "sh", 1.0, "config 5 at its quoted size: 10 M Gaussians, 4K, SH 3, forward (near / far split, band lists, "
                                         "difference-grid counting all switch on by size)"),
])
def test_full_size_frames_beside_the_reference_build_without_contraction(P, W, H, colors, scale_mult, label):
    """BASELINE configs 3 and 5 shapes against oracle/_ref's -ffp-contract=off build (the arithmetic the reference's sources spell,
    which the preprocess kernel restates): radii, num_rendered, n_contrib BIT-EXACT, final_T bit for bit, NO pixel over 1e-4 (image
    within 2e-6), config 3 also every gradient within 1e-3."""
    import torch
    from ref_mode_checks import need_ref
    ref_hip = need_ref("nofma")
    import wg_scenes as S
    from tests.wg_testlib import run_hip, run_hip_native
    deg = 3 if colors == "sh" else None
    d = deg if deg is not None else 0
    cloud = S.make_cloud(P, W, H, sh_degree=deg, seed=0, scale_mult=scale_mult)
    cam = S.make_camera(W, H)
    backward = colors == "precomp"
    cot = S.make_cotangent(W, H, seed=4) if backward else None
    r = ref_hip.run_scene(cloud, cam, sh_degree=d, cotangent=cot, variant="nofma")
    torch.cuda.empty_cache()
    h = run_hip(cloud, cam, sh_degree=d, cotangent=cot)
    n = run_hip_native(cloud, cam, sh_degree=d)
    assert int(n["num_rendered"]) == int(r["num_rendered"])
    assert np.array_equal(h["radii"], r["radii"])
    err = np.abs(h["color"].astype(np.float64) - r["color"]).max(axis=0)
    nflip = int((err > 1e-4).sum())
    ncon = n["views"]["image"]["n_contrib"].cpu().numpy().reshape(H, W).astype(np.int64)
    fT = n["views"]["image"]["final_T"].cpu().numpy().reshape(H, W)
    ncon_diff = int((ncon != r["n_contrib"].astype(np.int64)).sum())
    fT_diff = int((fT.view(np.uint32) != r["final_T"].astype(np.float32).view(np.uint32)).sum())
    print({"case": label, "num_rendered": int(r["num_rendered"]), "pixels_over_1e-4": nflip, "max": float(err.max()),
           "n_contrib_mismatch": ncon_diff, "final_T_bits_mismatch": fT_diff})
    # decision-exact compositing (Options::exact_compositing, the default): every skip / stop decision is the reference's, so
    # n_contrib and final_T are its bits and the image differs only by the colour sums' fused multiply-adds
    assert ncon_diff == 0 and fT_diff == 0, (ncon_diff, fT_diff)
    assert nflip == 0 and err.max() <= EXACT_IMG_ATOL, (nflip, float(err.max()))
    if backward:
        for k, g in h["grads"].items():
            assert _rel(g, r["grads"][k]) <= 1e-3, (k, _rel(g, r["grads"][k]))


@pytest.mark.gpu
def test_config4_cameras_beside_the_reference_build_without_contraction():
    """BASELINE config 4: the eight view-parallel cameras (wg_viewparallel.view_cameras: the base camera yawed by 0..35 degrees) over
    the 1 M-Gaussian headline cloud at 1080p, forward + backward each, beside oracle/_ref's -ffp-contract=off build: radii and
    num_rendered, n_contrib bit-exact, final_T bit for bit, NO pixel over 1e-4 (image within 2e-6), every gradient within 1e-3.  This is what each rank of `bench.py --gpus 8` computes."""
    import torch
    from ref_mode_checks import need_ref
    ref_hip = need_ref("nofma")
    import wg_scenes as S
    import wg_viewparallel as VP
    from tests.wg_testlib import run_hip, run_hip_native
    P, W, H = 1_000_000, 1920, 1080
    cloud = S.make_cloud(P, W, H, sh_degree=3, seed=0)
    cot = S.make_cotangent(W, H)
    report = []
    for k, cam in enumerate(VP.view_cameras(8, W, H)):
        r = ref_hip.run_scene(cloud, cam, sh_degree=3, cotangent=cot, variant="nofma")
        h = run_hip(cloud, cam, sh_degree=3, cotangent=cot)
        n = run_hip_native(cloud, cam, sh_degree=3)
        assert int(n["num_rendered"]) == int(r["num_rendered"]), k
        assert np.array_equal(h["radii"], r["radii"]), k
        err = np.abs(h["color"].astype(np.float64) - r["color"]).max(axis=0)
        nflip = int((err > 1e-4).sum())
        worst = max(_rel(g, r["grads"][key]) for key, g in h["grads"].items())
        ncon = n["views"]["image"]["n_contrib"].cpu().numpy().reshape(H, W).astype(np.int64)
        fT = n["views"]["image"]["final_T"].cpu().numpy().reshape(H, W)
        report.append({"view": k, "num_rendered": int(r["num_rendered"]), "pixels_over_1e-4": nflip, "max": float(err.max()), "grad_worst": worst,
                       "n_contrib_mismatch": int((ncon != r["n_contrib"].astype(np.int64)).sum()),
                       "final_T_bits_mismatch": int((fT.view(np.uint32) != r["final_T"].astype(np.float32).view(np.uint32)).sum())})
        assert report[-1]["n_contrib_mismatch"] == 0 and report[-1]["final_T_bits_mismatch"] == 0, report[-1]
        assert nflip == 0 and err.max() <= EXACT_IMG_ATOL, report[-1]
        for key, g in h["grads"].items():
            assert _rel(g, r["grads"][key]) <= 1e-3, (k, key, _rel(g, r["grads"][key]))
        del r, h, n
        torch.cuda.empty_cache()
    print(report)


# ---- full-size pins that need no reference binary -----------------------------------------------------------------------------------
import sys as _sys  # noqa: E402
_sys.path.insert(0, os.path.join(HERE, "golden"))
import fullsize_frames as FF  # noqa: E402


def test_full_size_pins_are_committed_for_every_frame():
    """tests/golden/ref_hip_fullsize_sha256.json (written by tests/golden/make_fullsize_ref_hashes.py on an MI355X from the reference's
    own kernels): one record per frame of tests/golden/fullsize_frames.py -- the headline, configs 2, 3, 5 and the config-4 cameras."""
    pins = json.load(open(FF.PINS))
    assert "nofma" in pins["reference_build"] and set(pins["frames"]) == set(FF.FRAMES)
    for name, rec in pins["frames"].items():
        assert rec["num_rendered"] > 0 and all(len(rec[k]) == 64 for k in ("radii_sha256", "n_contrib_sha256", "final_T_sha256")), name


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(FF.FRAMES))
def test_product_hashes_to_the_reference_pins_at_full_size(name):
    """The product's num_rendered, radii, n_contrib and final_T of every BASELINE frame at its quoted size hash to what the reference's
    own kernels (-ffp-contract=off build) gave on an MI355X: a full-size, bit-level parity pin that survives a clone without oracle/_ref."""
    import torch
    from tests.wg_testlib import run_hip_native
    pins = json.load(open(FF.PINS))["frames"][name]
    cloud, cam, deg = FF.frame_inputs(name)
    n = run_hip_native(cloud, cam, sh_degree=deg)
    im = n["views"]["image"]
    got = FF.digest(n["num_rendered"], n["radii"].cpu().numpy(), im["n_contrib"].cpu().numpy(), im["final_T"].cpu().numpy())
    del n, im
    torch.cuda.empty_cache()
    assert got == pins, {k: (got[k], pins[k]) for k in got if got[k] != pins[k]}
