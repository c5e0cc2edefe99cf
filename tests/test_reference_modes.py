"""Every call mode beyond the reference's surface, held to the REFERENCE'S OWN KERNELS inside the driver-run `-m gpu` suite
(VERDICT r4 "next" item 1).  The modes replace wildgaussians/method.py:1060-1086 (get_gaussians -> `filter_3D=`), :1573-1611 (the two
GaussianRasterizer calls of a step -> `colors_precomp2=` / `sh_second=`) and :890-900, 1590-1595 (the appearance toning -> `sh_mul=` /
`sh_offset=`); round 4 compared them in this suite with the product's own plain calls only, and the one real bug of that round (the
gradient-record clear of a two-colour backward stopped at the last whole float4: P = 1, 2, 3 mod 4 -> garbage dL_dcolor2) was found by
a sweep outside it.  Here each image is compared with its own run of oracle/_ref (the -ffp-contract=off build), the geometry gradients
with the sum of the two runs', the abs-gradient column separately -- checkers in tests/ref_mode_checks.py, shared with the long sweeps
of tests/tools/.  A box with a HIP device and without oracle/_ref FAILS these tests.

tests/tools/mutation_check_r43.sh re-introduces that bug in a variant build and shows this file going red on it."""
import numpy as np
import pytest
import torch

import ref_mode_checks as RC
import wg_scenes as S

pytestmark = pytest.mark.gpu

PAIR_CASES = [i for i in range(40) if i % 6 != 5]   # sweep cases with a scale / rotation pair


@pytest.fixture(scope="module")
def ref_hip():
    if not torch.cuda.is_available():
        pytest.fail("a HIP device is required for -m gpu tests (no CPU fallback exists)")
    return RC.need_ref("nofma")


def _assert_ok(rep):
    assert rep["ok"], rep


@pytest.mark.parametrize("i", range(16))
def test_two_colour_sets_in_one_call_beside_two_runs_of_the_reference(ref_hip, i):
    """P = 0, 1, 2, 3 (mod 4) four times over; cases 5 and 11 carry precomputed covariances."""
    _assert_ok(RC.check_two_colour(RC.mode_case(i, "precomp", pmod=i % 4), ref_hip, seed=i))


@pytest.mark.parametrize("path", RC.PATHS[1:])
@pytest.mark.parametrize("i", [1, 2, 3, 5])
def test_two_colour_sets_on_the_alternative_binning_paths(ref_hip, path, i):
    with RC.binning_path(path):
        _assert_ok(RC.check_two_colour(RC.mode_case(i, "precomp", pmod=i % 4), ref_hip, seed=100 + i))


@pytest.mark.parametrize("i", range(16))
def test_two_tones_of_one_sh_block_beside_two_runs_of_the_reference(ref_hip, i):
    """Every SH degree four times (degree = i % 4), P = (i // 4 + i) mod 4, every fourth case in WildGaussians' own shape (second set =
    the clamped coefficients alone); cases 5 and 11 carry precomputed covariances."""
    _assert_ok(RC.check_two_tone(RC.mode_case(i, "sh", pmod=(i // 4 + i) % 4, sh_degree=i % 4), ref_hip, seed=i, second_plain=i % 4 == 1))


@pytest.mark.parametrize("path", RC.PATHS[1:])
@pytest.mark.parametrize("i", [2, 3, 5, 9])
def test_two_tones_on_the_alternative_binning_paths(ref_hip, path, i):
    with RC.binning_path(path):
        _assert_ok(RC.check_two_tone(RC.mode_case(i, "sh", pmod=(i + 1) % 4, sh_degree=i % 4), ref_hip, seed=100 + i, second_plain=i == 9))


@pytest.mark.parametrize("i", range(12))
def test_one_tone_beside_the_reference_on_the_toned_coefficients(ref_hip, i):
    _assert_ok(RC.check_one_tone(RC.mode_case(i, "sh", pmod=(i + 2) % 4, sh_degree=i % 4), ref_hip, seed=i))


@pytest.mark.parametrize("i", PAIR_CASES[:12])
def test_raw_parameter_mode_beside_the_reference_on_the_activated_parameters(ref_hip, i):
    _assert_ok(RC.check_raw(RC.mode_case(i, "precomp", pmod=(i + 1) % 4, geometry="pair"), ref_hip, seed=i))


@pytest.mark.parametrize("path", RC.PATHS[1:])
def test_raw_parameter_mode_on_the_alternative_binning_paths(ref_hip, path):
    with RC.binning_path(path):
        _assert_ok(RC.check_raw(RC.mode_case(7, "precomp", pmod=3, geometry="pair"), ref_hip, seed=107))


def _config3_case(colours):
    """BASELINE config 3's size: 3 M Gaussians, 1600x1200 (P = 3 000 001 for the two-colour call: 1 mod 4)."""
    W, H = 1600, 1200
    P = 3_000_001 if colours == "precomp" else 3_000_000
    cloud = S.make_cloud(P, W, H, sh_degree=None if colours == "precomp" else 3, seed=0)
    return cloud, S.make_camera(W, H), (0 if colours == "precomp" else 3), dict(kernel_size=0.1, bg=None, subpixel_offset=None, scale_modifier=1.0), W, H


@pytest.mark.parametrize("mode", ["two_colour", "two_tone", "raw"])
def test_call_modes_at_config3_size_beside_the_reference(ref_hip, mode):
    """3 M Gaussians at 1600x1200 -- the size WildGaussians' step is quoted on (BASELINE config 3) -- through each mode, forward and
    backward, beside two (one) full runs of the reference's kernels."""
    torch.cuda.empty_cache()
    if mode == "two_colour":
        rep = RC.check_two_colour(_config3_case("precomp"), ref_hip, seed=3)
    elif mode == "two_tone":
        rep = RC.check_two_tone(_config3_case("sh"), ref_hip, seed=3, second_plain=True)
    else:
        rep = RC.check_raw(_config3_case("precomp"), ref_hip, seed=3)
    print(rep)
    _assert_ok(rep)
    torch.cuda.empty_cache()


def test_geometry_outside_the_reference_domain(ref_hip):
    """NaN / +-Inf / zero / negative means, scales, quaternions and opacities (tests/ref_mode_checks.py: NONFINITE_CATEGORIES), eight
    Gaussians of 20 000 each: in every category where the reference's output is DEFINED -- sixteen of eighteen, a NaN opacity included
    (alpha = fminf(0.99, NaN) = 0.99 over the Gaussian's whole tile rectangle: matched since round 6) -- the product does exactly what the
    reference's kernels do (radii and accumulation bit-identical, image within 1e-6).  In the other two (NONFINITE_REFERENCE_UNDEFINED: a NaN
    covariance gives radius 0 with one tile, whose key-list slot the reference never writes and then composites as stale memory) the radii
    and num_rendered agree and the product's image stays finite.  All eighteen at once -- on which the reference's own kernels end in a
    memory access fault (profiles/r5/nonfinite_inputs_ref.log) -- leave the product with a finite image and no fault
    (tests/tools/nonfinite_inputs.py prints the table)."""
    from wg_testlib import run_hip
    W, H, P = 640, 360, 20_000
    cam = S.make_camera(W, H)
    base = S.make_cloud(P, W, H, sh_degree=1, seed=11, scale_mult=2.0)
    rng = np.random.default_rng(3)
    every, untouched = base, np.ones(P, bool)
    for name in RC.NONFINITE_CATEGORIES:
        ids = rng.choice(P, size=8, replace=False)
        every = RC.poison(every, name, ids)
        untouched[ids] = False
        cloud = RC.poison(base, name, ids)
        h = run_hip(cloud, cam, sh_degree=1)
        r = ref_hip.run_scene(cloud, cam, sh_degree=1, variant="nofma")
        assert np.array_equal(h["radii"], r["radii"]), name
        assert np.isfinite(h["color"]).all() and np.isfinite(h["accumulation"]).all(), name
        if name not in RC.NONFINITE_DEVIATING:
            assert np.array_equal(h["accumulation"], r["accumulation"]), name
            assert float(np.abs(h["color"].astype(np.float64) - r["color"]).max()) <= 1e-6, name
    cot = S.make_cotangent(W, H)
    h = run_hip(every, cam, sh_degree=1, cotangent=cot)
    assert np.isfinite(h["color"]).all() and np.isfinite(h["accumulation"]).all()
    for k, g in h["grads"].items():   # the poisoned Gaussians' own gradients may be anything; nobody else's may be non-finite
        bad = ~np.isfinite(g.reshape(P, -1)[untouched]).all(axis=1)
        assert not bad.any(), (k, int(bad.sum()))
