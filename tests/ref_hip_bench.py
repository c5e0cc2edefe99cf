#!/usr/bin/env python3
"""The reference's OWN kernels on this GPU, beside ours: time oracle/_ref/libref_hip_rasterizer.so (hipcc build of the
reference's CUDA sources, oracle/ref_hip/) on a bench.py workload and compare its outputs with the product's at full size.

    python tests/ref_hip_bench.py [--gaussians 1000000 --width 1920 --height 1080 --colors sh --steps 20 --warmup 3]

Prints one JSON line.  bench.py runs it as a subprocess (with a timeout) inside its baseline leg; it is a checker and a
reported baseline, never part of the product path.  Needs a GPU and the prebuilt .so (no /root/reference at run time).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wild-gaussians_amd"))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gaussians", type=int, default=1_000_000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--colors", choices=["sh", "precomp"], default="sh")
    ap.add_argument("--scale-mult", type=float, default=1.0)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--variant", default="default")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--parity-variant", default=None, help="build of the reference the product's outputs are compared with (default: --variant); "
                    "'nofma' = -ffp-contract=off, the arithmetic the reference's sources spell and the product's decision-exact compositing restates")
    ap.add_argument("--forward-only", action="store_true", help="BASELINE config 5 style: time (and compare) the forward pass only")
    args = ap.parse_args()

    import wg_scenes as S
    from oracle.ref_hip import ref_hip
    from tests.wg_testlib import run_hip, run_hip_native, rel_err

    W, H, P = args.width, args.height, args.gaussians
    deg = 3 if args.colors == "sh" else None
    cloud = S.make_cloud(P, W, H, sh_degree=deg, seed=0, scale_mult=args.scale_mult)
    cam = S.make_camera(W, H)
    cot_np = S.make_cotangent(W, H)
    s = ref_hip.Session(cloud, cam, sh_degree=deg if deg is not None else 0, variant=args.variant)
    cot = torch.from_numpy(cot_np).cuda()

    def timed(fn, n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    def step():
        s.forward(copy_image_state=False)
        s.backward(cot)

    if args.forward_only:
        step = lambda: s.forward(copy_image_state=False)  # noqa: E731
    for _ in range(args.warmup):
        step()
    ms_step = timed(step, args.steps)
    ms_fwd = timed(lambda: s.forward(copy_image_state=False), args.steps)
    out = {"what": "reference CUDA sources compiled for gfx950 with hipcc (oracle/ref_hip), same workload, same GPU",
           "variant": args.variant, "gaussians": P, "width": W, "height": H, "colors": args.colors,
           "train_ms": round(ms_step, 3), "train_iters_per_s": round(1e3 / ms_step, 2),
           "forward_ms": round(ms_fwd, 3), "forward_fps": round(1e3 / ms_fwd, 2), "num_rendered": int(s.num_rendered)}
    if args.forward_only:
        out.pop("train_ms"), out.pop("train_iters_per_s")
    if not args.no_parity:
        pv = args.parity_variant or args.variant
        timed_build = None
        if pv != args.variant:
            # parity against BOTH builds of the reference.  The timed (compiler-default, contracting) build's forward outputs are
            # kept for a second comparison: what a user's default build of the reference would give on this GPU
            s.forward()
            torch.cuda.synchronize()
            timed_build = dict(color=s.color.cpu().numpy(), radii=s.radii.cpu().numpy(), n_contrib=s.n_contrib.cpu().numpy().reshape(H, W).astype(np.int64))
            ref_hip._lib(args.variant).refhip_release()
            del s
            torch.cuda.empty_cache()
            s = ref_hip.Session(cloud, cam, sh_degree=deg if deg is not None else 0, variant=pv)
        s.forward()
        if not args.forward_only:
            s.backward(cot)
        torch.cuda.synchronize()
        ref_color = s.color.cpu().numpy()
        ref_radii = s.radii.cpu().numpy()
        ref_T = s.final_T.cpu().numpy().reshape(H, W)
        ref_nc = s.n_contrib.cpu().numpy().reshape(H, W).astype(np.int64)
        ref_g = {} if args.forward_only else {k: v.cpu().numpy() for k, v in s.g.items()}
        ref_hip._lib(pv).refhip_release()
        del s
        torch.cuda.empty_cache()
        h = run_hip(cloud, cam, sh_degree=deg if deg is not None else 0, cotangent=None if args.forward_only else cot_np)
        h.setdefault("grads", {})
        err = np.abs(h["color"].astype(np.float64) - ref_color).max(axis=0)
        nv = run_hip_native(cloud, cam, sh_degree=deg if deg is not None else 0)["views"]["image"]
        out["product_vs_reference"] = {
            "reference_build": pv + (" (-ffp-contract=off)" if pv == "nofma" else " (compiler defaults: contraction on)"),
            "n_contrib_mismatch": int((nv["n_contrib"].cpu().numpy().reshape(H, W).astype(np.int64) != ref_nc).sum()),
            "final_T_bits_mismatch": int((nv["final_T"].cpu().numpy().reshape(H, W).view(np.uint32) != ref_T.astype(np.float32).view(np.uint32)).sum()),
            "color_max_abs": float(err.max()), "color_p9999_abs": float(np.quantile(err, 0.9999)),
            "pixels_over_1e-4": int((err > 1e-4).sum()), "pixels": int(err.size),
            "accumulation_max_abs": float(np.abs(h["accumulation"].reshape(H, W) - (1.0 - ref_T)).max()),
            "radii_mismatch": int((h["radii"] != ref_radii).sum()),
            "grad_max_rel_err": {k: float(f"{rel_err(g.reshape(ref_g[k].shape), ref_g[k]):.3e}") for k, g in h["grads"].items() if k in ref_g},
        }
        if ref_g:
            out["product_vs_reference"]["grad_max_rel_err_worst"] = max(out["product_vs_reference"]["grad_max_rel_err"].values())
        if timed_build is not None:
            e2 = np.abs(h["color"].astype(np.float64) - timed_build["color"]).max(axis=0)
            between = np.abs(ref_color.astype(np.float64) - timed_build["color"]).max(axis=0)
            out["product_vs_reference_default_build"] = {
                "reference_build": args.variant + " (compiler defaults: contraction on -- its fused multiply-adds move a radius or a threshold decision by an ulp here and there)",
                "pixels_over_1e-4": int((e2 > 1e-4).sum()), "color_max_abs": float(e2.max()),
                "radii_mismatch": int((h["radii"] != timed_build["radii"]).sum()),
                "n_contrib_mismatch": int((nv["n_contrib"].cpu().numpy().reshape(H, W).astype(np.int64) != timed_build["n_contrib"]).sum()),
                "the_reference's_two_builds_differ_from_each_other": {"pixels_over_1e-4": int((between > 1e-4).sum()),
                                                                      "radii": int((ref_radii != timed_build["radii"]).sum())},
                "pixels_over_1e-4_where_the_two_reference_builds_agree": int(((e2 > 1e-4) & ~(between > 1e-4)).sum())}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
