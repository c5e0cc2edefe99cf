"""How many (instance, strip) evaluations would an exact ellipse-vs-strip test save the render kernels?

CPU estimate on the bench scene through the oracle (test infrastructure): for every VISITED instance of every tile (list position
below the tile's largest n_contrib) and each of the tile's four 8x8 quadrants, three predicates:
  aabb   the current strip_mask(): the alpha >= 1/255 ellipse's axis-aligned box meets the quadrant's sample box (wg_alpha.h)
  exact  the minimum of the conic's quadratic form over the quadrant's sample box is <= 2 ln(255 o)
  hit    some pixel of the quadrant passes the reference's two skips (forward.cu:353-366), ignoring per-pixel termination
Prints the three per-instance averages and the fraction of visited instances without any reached strip.

    python tests/tools/strip_reach_estimate.py [--gaussians 1000000 --width 1920 --height 1080 --scale-mult 1.0 --tiles 600]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "wild-gaussians_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gaussians", type=int, default=1000000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--scale-mult", type=float, default=1.0)
    ap.add_argument("--tiles", type=int, default=600, help="random sample of tiles to evaluate")
    args = ap.parse_args()
    import wg_scenes as S
    from oracle import oracle as O

    W, H, P = args.width, args.height, args.gaussians
    cloud = S.make_cloud(P, W, H, sh_degree=None, seed=0, scale_mult=args.scale_mult)
    cam = S.make_camera(W, H)
    out = O.run_scene(cloud, cam, sh_degree=0)
    ctx = out["ctx"]
    m2 = ctx.get("means2D").astype(np.float64)
    co = ctx.get("conic_opacity").astype(np.float64)
    pl = ctx.get("point_list")
    ranges = ctx.get("ranges")
    ncontrib = ctx.get("n_contrib")
    gx, gy = (W + 15) // 16, (H + 15) // 16
    rng = np.random.default_rng(0)
    sample = rng.choice(gx * gy, size=min(args.tiles, gx * gy), replace=False)
    tot = dict(inst=0, aabb=0, exact=0, hit=0, none_aabb=0, none_exact=0, none_hit=0, listed=0)
    for t in sample:
        tx, ty = t % gx, t // gx
        x0, y0 = tx * 16, ty * 16
        nc = ncontrib[y0:y0 + 16, x0:x0 + 16]
        last = int(nc.max()) if nc.size else 0
        lo, hi = ranges[t]
        tot["listed"] += int(hi - lo)
        if last == 0:
            continue
        ids = pl[lo:lo + last]
        mx, my = m2[ids, 0], m2[ids, 1]
        A, B, Cc, o = co[ids, 0], co[ids, 1], co[ids, 2], co[ids, 3]
        tau2 = 2.0 * np.log(np.maximum(255.0 * o, 1e-30))
        det = A * Cc - B * B
        ex = np.sqrt(np.maximum(tau2 * Cc / det, 0.0))
        ey = np.sqrt(np.maximum(tau2 * A / det, 0.0))
        vis = 255.0 * o >= 1.0
        n = len(ids)
        m_aabb = np.zeros((n, 4), bool)
        m_exact = np.zeros((n, 4), bool)
        m_hit = np.zeros((n, 4), bool)
        for s in range(4):
            qx0, qy0 = x0 + 8 * (s & 1), y0 + 8 * (s >> 1)
            qx1, qy1 = min(qx0 + 7, W - 1), min(qy0 + 7, H - 1)
            if qx0 >= W or qy0 >= H:
                continue
            m_aabb[:, s] = vis & (mx + ex >= qx0) & (mx - ex <= qx1) & (my + ey >= qy0) & (my - ey <= qy1)
            # exact: min over the box of A dx^2 + 2 B dx dy + C dy^2, d = p - mean
            X0, X1, Y0, Y1 = qx0 - mx, qx1 - mx, qy0 - my, qy1 - my
            inside = (X0 <= 0) & (X1 >= 0) & (Y0 <= 0) & (Y1 >= 0)
            best = np.full(n, np.inf)
            for X in (X0, X1):
                dy = np.clip(-B * X / Cc, Y0, Y1)
                best = np.minimum(best, A * X * X + 2 * B * X * dy + Cc * dy * dy)
            for Y in (Y0, Y1):
                dx = np.clip(-B * Y / A, X0, X1)
                best = np.minimum(best, A * dx * dx + 2 * B * dx * Y + Cc * Y * Y)
            best = np.where(inside, 0.0, best)
            m_exact[:, s] = vis & (best <= tau2)
            px = np.arange(qx0, qx1 + 1, dtype=np.float64)
            py = np.arange(qy0, qy1 + 1, dtype=np.float64)
            dx = mx[:, None, None] - px[None, None, :]
            dy = my[:, None, None] - py[None, :, None]
            power = -0.5 * (A[:, None, None] * dx * dx + Cc[:, None, None] * dy * dy) - B[:, None, None] * dx * dy
            alpha = np.minimum(0.99, o[:, None, None] * np.exp(power))
            m_hit[:, s] = ((power <= 0) & (alpha >= 1.0 / 255.0)).any(axis=(1, 2))
        assert not (m_hit & ~m_exact).any(), "the exact test excluded a passing pixel"
        assert not (m_exact & ~m_aabb).any()
        tot["inst"] += n
        for k, m in (("exact", m_exact),):
            cnt = m.sum(axis=1)
            for c in range(5):
                tot[f"n{c}"] = tot.get(f"n{c}", 0) + int((cnt == c).sum())
            tot["pairs_rows"] = tot.get("pairs_rows", 0) + int((m[:, 0] | m[:, 1]).sum() + (m[:, 2] | m[:, 3]).sum())   # top / bottom halves
            tot["pairs_cols"] = tot.get("pairs_cols", 0) + int((m[:, 0] | m[:, 2]).sum() + (m[:, 1] | m[:, 3]).sum())   # left / right halves
            tot["both_rows"] = tot.get("both_rows", 0) + int((m[:, 0] & m[:, 1]).sum() + (m[:, 2] & m[:, 3]).sum())
        for k, m in (("aabb", m_aabb), ("exact", m_exact), ("hit", m_hit)):
            tot[k] += int(m.sum())
            tot["none_" + k] += int((~m.any(axis=1)).sum())
    n = tot["inst"]
    print(f"tiles sampled {len(sample)}, listed instances {tot['listed']}, visited {n} ({n / max(tot['listed'], 1):.3f})")
    ent = n - tot["n0"]
    print("exact: reached-strip count distribution over entering instances:", {c: round(tot[f"n{c}"] / ent, 3) for c in range(1, 5)})
    print(f"exact: strips / entering instance {tot['exact'] / ent:.3f}; strip PAIRS / entering instance: top|bottom halves "
          f"{tot['pairs_rows'] / ent:.3f} (both strips of the pair reached in {tot['both_rows'] / max(tot['pairs_rows'], 1):.3f} of them), "
          f"left|right halves {tot['pairs_cols'] / ent:.3f}")
    for k in ("aabb", "exact", "hit"):
        print(f"{k:6s} strips / visited instance {tot[k] / n:.3f}   instances with no strip {tot['none_' + k] / n:.3f}")


if __name__ == "__main__":
    main()
