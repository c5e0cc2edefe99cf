#!/usr/bin/env python3
"""Generate tests/golden/ref_hip_golden.npz: outputs of the REFERENCE ITSELF for the cases of ref_hip_cases.py.

The reference's rasterizer is CUDA only; oracle/ref_hip/Makefile compiles those same sources (where they lie under
/root/reference) for gfx950 with hipcc -- the build container does that, the .so travels -- and this script runs the
result ON A GPU BOX:

    gpurun -- 'python tests/golden/make_golden_ref_hip.py'        # writes gpurun_out/ref_hip_golden.npz (+ a report)
    cp gpurun_out/ref_hip_golden.npz tests/golden/                  # back in the build container, then commit

Two builds are recorded: "default" (compiler-default fp contraction, like nvcc) and "nofma" (-ffp-contract=off: the
arithmetic the sources spell, which is what oracle/wg_oracle.c restates).  The report compares the CPU oracle and the HIP
product with both, which is how the tolerances in tests/test_reference_golden.py were chosen.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "wild-gaussians_amd"))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

OUT_KEYS = ["color", "radii", "final_T", "n_contrib", "visible"]


def main():
    import wg_scenes as S
    from oracle import oracle
    from oracle.ref_hip import ref_hip
    from ref_hip_cases import cases
    from tests.wg_testlib import run_hip, rel_err

    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    store, report = {}, {}
    names = []
    for name, cloud, cam, kw in cases():
        names.append(name)
        cot = S.make_cotangent(cam["width"], cam["height"], seed=len(names))
        for k, v in cloud.items():
            store[f"{name}/in/{k}"] = v
        for k in ("viewmatrix", "projmatrix", "campos"):
            store[f"{name}/cam/{k}"] = np.asarray(cam[k], np.float32)
        store[f"{name}/cam/scalars"] = np.array([cam["width"], cam["height"], cam["tanfovx"], cam["tanfovy"]], np.float64)
        store[f"{name}/in/cotangent"] = cot
        store[f"{name}/kw"] = np.array(json.dumps({k: (v.tolist() if isinstance(v, np.ndarray) else v)
                                                   for k, v in kw.items() if k != "subpixel_offset"}))
        if kw.get("subpixel_offset") is not None:
            store[f"{name}/in/subpixel_offset"] = kw["subpixel_offset"]
        res = {}
        for variant in ("default", "nofma"):
            r = ref_hip.run_scene(cloud, cam, cotangent=cot, variant=variant, **kw)
            res[variant] = r
            store[f"{name}/{variant}/num_rendered"] = np.array(r["num_rendered"], np.int64)
            for k in OUT_KEYS:
                store[f"{name}/{variant}/{k}"] = r[k]
            for k, g in r["grads"].items():
                store[f"{name}/{variant}/grad/{k}"] = g
        o = oracle.run_scene(cloud, cam, cotangent=cot, **kw)
        o["final_T"] = np.asarray(o["ctx"].get("final_T")).reshape(cam["height"], cam["width"])
        o["n_contrib"] = np.asarray(o["ctx"].get("n_contrib")).reshape(cam["height"], cam["width"])
        h = run_hip(cloud, cam, cotangent=cot, **kw)
        h["grads"] = dict(h["grads"])
        rep = {}
        for who, x in (("oracle", o), ("product", h)):
            for variant in ("default", "nofma"):
                r = res[variant]
                d = {"color_max_abs": float(np.abs(x["color"] - r["color"]).max()),
                     "color_pixels_over_1e-4": int((np.abs(x["color"] - r["color"]).max(axis=0) > 1e-4).sum()),
                     "radii_mismatch": int((x["radii"] != r["radii"]).sum())}
                if "num_rendered" in x:
                    d["num_rendered"] = [int(x["num_rendered"]), int(r["num_rendered"])]
                if "n_contrib" in x:
                    d["n_contrib_mismatch"] = int((x["n_contrib"] != r["n_contrib"]).sum())
                    d["final_T_max_abs"] = float(np.abs(x["final_T"] - r["final_T"]).max())
                else:
                    d["accumulation_max_abs"] = float(np.abs(x["accumulation"].reshape(r["accumulation"].shape) - r["accumulation"]).max())
                d["grads"] = {k: rel_err(np.asarray(x["grads"][k]).reshape(g.shape), g) for k, g in r["grads"].items()
                              if k in x["grads"] and g.size}
                rep[f"{who}_vs_{variant}"] = d
        rep["default_vs_nofma"] = {"color_max_abs": float(np.abs(res["default"]["color"] - res["nofma"]["color"]).max()),
                                   "radii_mismatch": int((res["default"]["radii"] != res["nofma"]["radii"]).sum()),
                                   "grads": {k: rel_err(res["default"]["grads"][k], g) for k, g in res["nofma"]["grads"].items() if g.size}}
        rep["stats"] = {"num_rendered": int(res["default"]["num_rendered"]), "visible": int((res["default"]["radii"] > 0).sum()),
                        "n_contrib_max": int(res["default"]["n_contrib"].max())}
        report[name] = rep
        print(name, json.dumps(rep["stats"]), flush=True)
    # the reference's simple-knn itself on the point sets of tests/test_knn.py
    if ref_hip.knn_available():
        import torch
        from simple_knn._C import distCUDA2
        from tests.test_knn import _clouds
        knn_store, knn_report = {}, {}
        for name, pts in _clouds().items():
            ref = ref_hip.dist_cuda2(pts)
            ref_nofma = ref_hip.dist_cuda2(pts, variant="nofma")
            knn_store[f"{name}/points"], knn_store[f"{name}/mean_dist2"], knn_store[f"{name}/mean_dist2_nofma"] = pts, ref, ref_nofma
            o = oracle.dist_cuda2(pts)
            h = distCUDA2(torch.from_numpy(pts).cuda()).cpu().numpy()
            knn_report[name] = {"P": int(pts.shape[0]), "oracle_bit_mismatches": int((o.view(np.uint32) != ref.view(np.uint32)).sum()),
                                "product_bit_mismatches": int((h.view(np.uint32) != ref.view(np.uint32)).sum()),
                                "oracle_bit_mismatches_nofma": int((o.view(np.uint32) != ref_nofma.view(np.uint32)).sum()),
                                "product_bit_mismatches_nofma": int((h.view(np.uint32) != ref_nofma.view(np.uint32)).sum()),
                                "oracle_max_rel": float(np.nanmax(np.abs(o - ref) / (np.abs(ref) + 1e-30))) if np.isfinite(ref).all() else None}
        knn_store["names"] = np.array(list(_clouds()))
        np.savez_compressed(os.path.join(out_dir, "ref_hip_knn_golden.npz"), **knn_store)
        report["simple_knn"] = knn_report
        print("simple_knn", json.dumps(knn_report), flush=True)
    store["names"] = np.array(names)
    np.savez_compressed(os.path.join(out_dir, "ref_hip_golden.npz"), **store)
    with open(os.path.join(out_dir, "ref_hip_golden_report.json"), "w") as f:
        json.dump(report, f, indent=1)
    print(json.dumps(report, indent=1))


if __name__ == "__main__":
    main()
