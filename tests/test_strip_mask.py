"""The render kernels' exact ellipse-vs-strip reach test (csrc/wg_alpha.h: strip_mask_exact) must be conservative: a strip it
excludes holds no pixel that passes the reference's alpha >= 1/255 skip (forward.cu:353-366).  CPU check of the arithmetic: a
float32 numpy mirror of the device function, operation for operation, against the per-pixel float32 evaluation the kernels perform
(csrc/wg_alpha.h: eval_alpha), over conics from round to needle-thin and strongly correlated, tiles near and far from the mean.
(The device function itself is covered by the GPU parity tests: an excluded passing pixel would change n_contrib and the image.)"""
import numpy as np

f32 = np.float32
LOG2E = f32(1.4426950408889634)


def med3(a, b, c):
    return np.maximum(np.minimum(a, b), np.minimum(np.maximum(a, b), c))


def strip_mask_exact_mirror(mx, my, A, B, C, o, sb):
    """sb: four (x0, x1, y0, y1) sample boxes.  All float32 arrays of one length; returns [n, 4] bool."""
    with np.errstate(all="ignore"):
        vis = o * f32(1.001) >= f32(1.0 / 255.0)
        AC = A * C
        definite = (A > 0) & (C > 0) & ((AC - B * B) > f32(4e-6) * AC)
        tau2 = f32(2.0) * f32(0.6931471805599453) * np.log2(f32(255.0) * o).astype(f32) * f32(1.002) + f32(0.002)
        nBC, nBA, B2 = -B * (f32(1) / C), -B * (f32(1) / A), f32(2) * B
        Mx = np.zeros_like(mx)
        My = np.zeros_like(mx)
        for x0, x1, y0, y1 in sb:
            Mx = np.maximum(Mx, np.maximum(np.abs(f32(x0) - mx), np.abs(f32(x1) - mx)))
            My = np.maximum(My, np.maximum(np.abs(f32(y0) - my), np.abs(f32(y1) - my)))
        thr = tau2 + f32(2e-6) * (A * Mx * Mx + np.abs(B2) * Mx * My + C * My * My)
        m = np.zeros((len(mx), 4), bool)
        for s, (x0, x1, y0, y1) in enumerate(sb):
            X0, X1, Y0, Y1 = f32(x0) - mx, f32(x1) - mx, f32(y0) - my, f32(y1) - my
            xf, yf = med3(X0, f32(0), X1), med3(Y0, f32(0), Y1)
            t = med3(nBC * xf, Y0, Y1)
            fv = xf * (A * xf + B2 * t) + C * t * t
            u = med3(nBA * yf, X0, X1)
            fh = yf * (C * yf + B2 * u) + A * u * u
            m[:, s] = np.minimum(fv, fh) <= thr
        return np.where(vis[:, None], np.where(definite[:, None], m, True), False)


def pixel_pass(mx, my, A, B, C, o, x0, y0):
    """Any pixel of the 8x8 quadrant at (x0, y0) passes eval_alpha(), in float32."""
    with np.errstate(all="ignore"):
        ca, cb, cc = f32(-0.5) * LOG2E * A, -LOG2E * B, f32(-0.5) * LOG2E * C
        px = (np.arange(8, dtype=f32) + f32(x0))[None, None, :]
        py = (np.arange(8, dtype=f32) + f32(y0))[None, :, None]
        dx, dy = mx[:, None, None] - px, my[:, None, None] - py
        p2 = ca[:, None, None] * (dx * dx) + cb[:, None, None] * (dx * dy) + cc[:, None, None] * (dy * dy)
        alpha = np.minimum(f32(0.99), o[:, None, None] * np.exp2(p2).astype(f32))
        return ((p2 <= 0) & (alpha >= f32(1.0 / 255.0))).any(axis=(1, 2))


def conics(rng, n, max_sigma):
    """Conics of 2-D covariances with the reference's +0.3 dilation: sigma_major up to max_sigma px, any orientation."""
    s1 = np.exp(rng.uniform(np.log(0.05), np.log(max_sigma), n))
    s2 = np.exp(rng.uniform(np.log(0.05), np.log(max_sigma), n))
    th = rng.uniform(0, np.pi, n)
    c, s = np.cos(th), np.sin(th)
    a = c * c * s1 * s1 + s * s * s2 * s2 + 0.3
    b = c * s * (s1 * s1 - s2 * s2)
    d = s * s * s1 * s1 + c * c * s2 * s2 + 0.3
    det = (a * d - b * b)
    return (d / det).astype(f32), (-b / det).astype(f32), (a / det).astype(f32)


def test_exact_strip_mask_never_excludes_a_passing_pixel():
    rng = np.random.default_rng(5)
    total_excluded = total = 0
    for max_sigma, spread in ((6.0, 24.0), (60.0, 120.0), (3000.0, 3000.0), (3000.0, 40.0)):
        n = 60000
        A, B, C = conics(rng, n, max_sigma)
        o = rng.uniform(0.003, 1.0, n).astype(f32)
        x0, y0 = 640, 352   # a tile well inside a 1080p frame; the mean is scattered around it
        mx = (x0 + 8 + rng.normal(0, spread, n)).astype(f32)
        my = (y0 + 8 + rng.normal(0, spread, n)).astype(f32)
        sb = [(x0 + 8 * (s & 1), x0 + 8 * (s & 1) + 7, y0 + 8 * (s >> 1), y0 + 8 * (s >> 1) + 7) for s in range(4)]
        m = strip_mask_exact_mirror(mx, my, A, B, C, o, sb)
        for s, (qx0, _, qy0, _) in enumerate(sb):
            hit = pixel_pass(mx, my, A, B, C, o, qx0, qy0)
            assert not (hit & ~m[:, s]).any(), f"strip {s}: the mask excluded a strip with a passing pixel (sigma <= {max_sigma})"
            total += int(hit.sum())
            total_excluded += int((~m[:, s]).sum())
    assert total > 20000 and total_excluded > 100000   # the cases exercise both answers


def test_exact_strip_mask_is_tight_on_the_pixel_grid():
    """With samples on the integer grid and moderate footprints the exact mask admits only a few percent more strips than hold a passing pixel."""
    rng = np.random.default_rng(6)
    n = 80000
    A, B, C = conics(rng, n, 6.0)
    o = rng.uniform(0.05, 0.95, n).astype(f32)
    x0, y0 = 640, 352
    mx = (x0 + 8 + rng.normal(0, 16.0, n)).astype(f32)
    my = (y0 + 8 + rng.normal(0, 16.0, n)).astype(f32)
    sb = [(x0 + 8 * (s & 1), x0 + 8 * (s & 1) + 7, y0 + 8 * (s >> 1), y0 + 8 * (s >> 1) + 7) for s in range(4)]
    m = strip_mask_exact_mirror(mx, my, A, B, C, o, sb)
    hits = sum(int(pixel_pass(mx, my, A, B, C, o, q[0], q[2]).sum()) for q in sb)
    assert hits <= int(m.sum()) <= 1.05 * hits
