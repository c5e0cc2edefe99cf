#!/usr/bin/env python3
"""The argument sweep of tests/test_parity_gpu.py over the two-tone call (sh_second=; wg_rasterize_*_two_tone), with the REFERENCE'S OWN
KERNELS (oracle/_ref, -ffp-contract=off) as the checker, on the sweep's SH cases (every degree, both coefficient layouts, scale /
rotation pairs and precomputed covariances).  The reference knows no tone: it is fed the toned coefficient tensors, built on the host with
the arithmetic the tone is defined by (float32 min, one multiply, one add, min), once per tone; its dL/dsh of each run is chained through
that tone on the host (clamp_max passes the gradient where x <= max).  Checked: each image against the reference's run with that tone's
coefficients; dL_dsh against the SUM of the two chained gradients, each tone's dL_dmul / dL_doffset against its run's, the geometry
gradients against the sum of the two runs'.
usage: python tests/tools/stress_sweep_two_tone.py [first] [count]   -> one summary line (and one line per deviation)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "wild-gaussians_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import wg_scenes as S
from oracle.ref_hip import ref_hip
from diff_gaussian_rasterization import GaussianRasterizer
from tests.wg_testlib import make_settings, to_dev, rel_err
from tests.test_parity_gpu import _sweep_case


def tone_host(sh, mul, off, pre, post):
    """(toned coefficients, chain) -- chain(dL/dtoned) -> (dL/dsh, dL/dmul, dL/doffset)"""
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


first, count = int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 500
st = dict(cases=0, pixels=0, pixels_over=0, radii_mismatch_runs=0, worst_img=0.0, worst_grad=0.0, worst_grad_name="")
for i in range(first, first + count):
    cloud, cam, deg, kw, W, H = _sweep_case(i)
    if "shs" not in cloud:
        continue
    st["cases"] += 1
    P, M = cloud["shs"].shape[:2]
    rng = np.random.default_rng(11000 + i)
    sh = (cloud["shs"] * np.float32(rng.choice([1.0, 3.0]))).astype(np.float32)
    mul1, off1 = rng.uniform(0.5, 1.5, size=(P, 3)).astype(np.float32), rng.normal(0, 0.3, size=(P, 3)).astype(np.float32)
    mul2, off2 = rng.uniform(0.8, 1.2, size=(P, 3)).astype(np.float32), rng.normal(0, 0.1, size=(P, 3)).astype(np.float32)
    pre1, post1, pre2, post2 = 1.0, 1.0, float(rng.choice([1.0, 0.3])), float(rng.choice([np.inf, 0.5]))
    second_plain = i % 4 == 1   # WildGaussians' shape: the second set is the clamped coefficients alone
    if second_plain:
        mul2, off2, post2 = np.ones_like(mul2), np.zeros_like(off2), np.inf
    sh1, chain1 = tone_host(sh, mul1, off1, pre1, post1)
    sh2, chain2 = tone_host(sh, mul2, off2, pre2, post2)
    cot1, cot2 = S.make_cotangent(W, H, seed=3000 + i), S.make_cotangent(W, H, seed=5000 + i)
    r1 = ref_hip.run_scene(dict(cloud, shs=sh1), cam, sh_degree=deg, cotangent=cot1, variant="nofma", **kw)
    r2 = ref_hip.run_scene(dict(cloud, shs=sh2), cam, sh_degree=deg, cotangent=cot2, variant="nofma", **kw)
    rs = make_settings(cam, deg, kw["kernel_size"], kw["bg"], kw["subpixel_offset"], kw["scale_modifier"])
    t = {k: to_dev(v).requires_grad_(True) for k, v in dict(cloud, shs=sh).items()}
    tn = dict(mul1=to_dev(mul1).requires_grad_(True), off1=to_dev(off1).requires_grad_(True))
    if not second_plain:
        tn.update(mul2=to_dev(mul2).requires_grad_(True), off2=to_dev(off2).requires_grad_(True))
    m2d = torch.zeros((P, 3), device="cuda", requires_grad=True)
    geo = {k: t[k] for k in ("scales", "rotations") if k in t}
    if "cov3D_precomp" in t:
        geo["cov3D_precomp"] = t["cov3D_precomp"]
    second = dict(sh_pre_clamp_max2=pre2) if second_plain else dict(sh_mul2=tn["mul2"], sh_offset2=tn["off2"], sh_pre_clamp_max2=pre2,
                                                                    sh_post_clamp_max2=None if post2 == np.inf else post2)
    img1, radii, acc, img2 = GaussianRasterizer(rs)(means3D=t["means3D"], means2D=m2d, opacities=t["opacities"], shs=t["shs"], **geo,
                                                   sh_mul=tn["mul1"], sh_offset=tn["off1"], sh_pre_clamp_max=pre1, sh_post_clamp_max=post1,
                                                   sh_second=True, **second)
    ((img1 * to_dev(cot1)).sum() + (img2 * to_dev(cot2)).sum()).backward()
    a1, a2 = img1.detach().cpu().numpy(), img2.detach().cpu().numpy()
    over = int((np.abs(a1 - r1["color"]).max(axis=0) > 1e-4).sum() + (np.abs(a2 - r2["color"]).max(axis=0) > 1e-4).sum())
    st["pixels_over"] += over
    st["pixels"] += 2 * W * H
    st["worst_img"] = max(st["worst_img"], float(np.abs(a1 - r1["color"]).max()), float(np.abs(a2 - r2["color"]).max()))
    st["radii_mismatch_runs"] += int(not np.array_equal(radii.cpu().numpy(), r1["radii"]))
    d1, dm1, do1 = chain1(r1["grads"]["sh"])
    d2, dm2, do2 = chain2(r2["grads"]["sh"])
    g = dict(sh=rel_err(t["shs"].grad.cpu().numpy(), (d1 + d2).astype(np.float32)), mul1=rel_err(tn["mul1"].grad.cpu().numpy(), dm1.astype(np.float32)),
             off1=rel_err(tn["off1"].grad.cpu().numpy(), do1.astype(np.float32)))
    if not second_plain:
        g.update(mul2=rel_err(tn["mul2"].grad.cpu().numpy(), dm2.astype(np.float32)), off2=rel_err(tn["off2"].grad.cpu().numpy(), do2.astype(np.float32)))
    for k, kk in (("means3D", "means3D"), ("opacities", "opacities"), ("scales", "scales"), ("rotations", "rotations"), ("cov3D_precomp", "cov3Ds_precomp")):
        if k in t:
            g[k] = rel_err(t[k].grad.cpu().numpy().reshape(r1["grads"][kk].shape), r1["grads"][kk] + r2["grads"][kk])
    g["means2D"] = rel_err(m2d.grad.cpu().numpy(), r1["grads"]["means2D"] + r2["grads"]["means2D"])
    worst = max(g, key=g.get)
    if g[worst] > st["worst_grad"]:
        st["worst_grad"], st["worst_grad_name"] = g[worst], worst
    if over or g[worst] > 1e-3:
        print("two-tone deviation: case", i, "deg", deg, "M", M, "pixels over", over, "worst gradient", worst, g[worst])
print(f"cases {first}..{first + count - 1}: SH colours:", st)
