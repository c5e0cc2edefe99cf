#!/usr/bin/env python3
"""The argument sweep of tests/test_parity_gpu.py over the two call modes round 4 added, with the REFERENCE'S OWN KERNELS (oracle/_ref,
-ffp-contract=off) as the checker, on the sweep's precomputed-colour cases with a scale / rotation pair:
  * two colour sets in ONE call (colors_precomp2=): each image against the reference's run with that colour set; each colour set's
    gradient against that run's; the geometry gradients against the SUM of the two runs' (what autograd adds over two calls);
  * raw-parameter mode (filter_3D=): the reference fed with the activated parameters (this repo's stand-alone activation kernels,
    wg_fused_gaussians.activate -- held to the PyTorch restatement of get_gaussians, method.py:1060-1086, by tests/test_activations.py;
    torch's own exp / sigmoid differ from them in last bits, which in these tiny scenes moves a decision now and then); gradients of the
    raw parameters against the reference's gradients chained through the activation kernels' backward pass.
usage: python tests/tools/stress_sweep_round4_modes.py [first] [count]   -> one summary line (and one line per deviation)"""
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
import wg_fused_gaussians as FG

first, count = int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 500
st = dict(cases=0, two_colour_pixels_over=0, two_colour_worst_img=0.0, two_colour_worst_grad=0.0, raw_radii_mismatch_runs=0, raw_pixels_over=0,
          raw_worst_img_p9999=0.0, raw_worst_grad=0.0, pixels=0)
for i in range(first, first + count):
    cloud, cam, deg, kw, W, H = _sweep_case(i)
    if "colors_precomp" not in cloud or "scales" not in cloud:
        continue
    st["cases"] += 1
    P = cloud["means3D"].shape[0]
    rng = np.random.default_rng(7000 + i)
    c2 = rng.uniform(0, 1, size=(P, 3)).astype(np.float32)
    cot1, cot2 = S.make_cotangent(W, H, seed=3000 + i), S.make_cotangent(W, H, seed=5000 + i)
    r1 = ref_hip.run_scene(cloud, cam, sh_degree=0, cotangent=cot1, variant="nofma", **kw)
    r2 = ref_hip.run_scene(dict(cloud, colors_precomp=c2), cam, sh_degree=0, cotangent=cot2, variant="nofma", **kw)
    rs = make_settings(cam, 0, kw["kernel_size"], kw["bg"], kw["subpixel_offset"], kw["scale_modifier"])
    t = {k: to_dev(v).requires_grad_(True) for k, v in cloud.items()}
    tc2 = to_dev(c2).requires_grad_(True)
    m2d = torch.zeros((P, 3), device="cuda", requires_grad=True)
    img1, radii, acc, img2 = GaussianRasterizer(rs)(means3D=t["means3D"], means2D=m2d, opacities=t["opacities"], scales=t["scales"], rotations=t["rotations"],
                                                   colors_precomp=t["colors_precomp"], colors_precomp2=tc2)
    ((img1 * to_dev(cot1)).sum() + (img2 * to_dev(cot2)).sum()).backward()
    e = max(float(np.abs(img1.detach().cpu().numpy() - r1["color"]).max()), float(np.abs(img2.detach().cpu().numpy() - r2["color"]).max()))
    over = int((np.abs(img1.detach().cpu().numpy() - r1["color"]).max(axis=0) > 1e-4).sum() + (np.abs(img2.detach().cpu().numpy() - r2["color"]).max(axis=0) > 1e-4).sum())
    st["two_colour_pixels_over"] += over
    st["pixels"] += 2 * W * H
    st["two_colour_worst_img"] = max(st["two_colour_worst_img"], e)
    g = dict(colors1=rel_err(t["colors_precomp"].grad.cpu().numpy(), r1["grads"]["colors_precomp"]), colors2=rel_err(tc2.grad.cpu().numpy(), r2["grads"]["colors_precomp"]))
    for k, kk in (("means3D", "means3D"), ("opacities", "opacities"), ("scales", "scales"), ("rotations", "rotations")):
        g[k] = rel_err(t[k].grad.cpu().numpy().reshape(r1["grads"][kk].shape), r1["grads"][kk] + r2["grads"][kk])
    g["means2D"] = rel_err(m2d.grad.cpu().numpy(), r1["grads"]["means2D"] + r2["grads"]["means2D"])
    st["two_colour_worst_grad"] = max(st["two_colour_worst_grad"], max(g.values()))
    if over or max(g.values()) > 1e-3 or not np.array_equal(radii.cpu().numpy(), r1["radii"]):
        print("two-colour deviation: case", i, "pixels over", over, "worst gradient", max(g, key=g.get), max(g.values()))
    # ---- raw-parameter mode: raw parameters whose activations are the case's cloud before the filter
    gen = torch.Generator().manual_seed(9000 + i)
    filt = (0.3 * torch.rand(P, 1, generator=gen) * torch.from_numpy(cloud["scales"]).mean(dim=1, keepdim=True)).cuda()
    raw = dict(opacities=torch.special.logit(torch.from_numpy(cloud["opacities"]).clamp(1e-4, 1 - 1e-4)).cuda().requires_grad_(True),
               scales=torch.log(torch.from_numpy(cloud["scales"])).cuda().requires_grad_(True),
               rotations=(torch.from_numpy(cloud["rotations"]) * (0.5 + torch.rand(P, 1, generator=gen))).cuda().requires_grad_(True))
    o, s_, q = FG.activate(raw["opacities"], raw["scales"], raw["rotations"], filt)
    rr = ref_hip.run_scene(dict(cloud, opacities=o.detach().cpu().numpy(), scales=s_.detach().cpu().numpy(), rotations=q.detach().cpu().numpy()), cam,
                           sh_degree=0, cotangent=cot1, variant="nofma", **kw)
    torch.autograd.backward([o, s_, q], [to_dev(rr["grads"]["opacities"]).view_as(o), to_dev(rr["grads"]["scales"]), to_dev(rr["grads"]["rotations"])])
    want = {k: v.grad.clone() for k, v in raw.items()}
    for v in raw.values():
        v.grad = None
    m3 = to_dev(cloud["means3D"]).requires_grad_(True)
    col = to_dev(cloud["colors_precomp"]).requires_grad_(True)
    img, radii, acc = GaussianRasterizer(rs)(means3D=m3, means2D=torch.zeros((P, 3), device="cuda", requires_grad=True), opacities=raw["opacities"],
                                            scales=raw["scales"], rotations=raw["rotations"], colors_precomp=col, filter_3D=filt)
    (img * to_dev(cot1)).sum().backward()
    err = np.abs(img.detach().cpu().numpy() - rr["color"]).max(axis=0)
    st["raw_radii_mismatch_runs"] += int(not np.array_equal(radii.cpu().numpy(), rr["radii"]))
    st["raw_pixels_over"] += int((err > 1e-4).sum())
    st["raw_worst_img_p9999"] = max(st["raw_worst_img_p9999"], float(np.quantile(err, 0.9999)))
    gr = {k: rel_err(raw[k].grad.cpu().numpy(), want[k].cpu().numpy()) for k in raw}
    gr["means3D"] = rel_err(m3.grad.cpu().numpy(), rr["grads"]["means3D"])
    st["raw_worst_grad"] = max(st["raw_worst_grad"], max(gr.values()))
    if max(gr.values()) > 1e-3:
        print("raw-parameter gradient over 1e-3: case", i, max(gr, key=gr.get), max(gr.values()))
print(f"cases {first}..{first + count - 1}: precomputed colours + scale / rotation pair:", st)
