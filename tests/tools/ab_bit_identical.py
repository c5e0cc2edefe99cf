"""Two builds of the library must give bit-identical results: a check for changes that may only skip work, never change it
(e.g. the render kernels' strip culling, wg_alpha.h: a conservative mask leaves every output bit as it was).

    python tests/tools/ab_bit_identical.py <libA.so> <libB.so>      (on the GPU box; libraries as wild-gaussians_amd/build.py
                                                                       variant builds leave them under wild-gaussians_amd/build/<name>/)

Each library renders the same scenes in its own process (WG_RASTERIZER_LIB selects it): the parity suite's cases, a needle scene,
the bench scene at 1080p and a dense variant, forward and -- in the deterministic backward mode, which is bit-reproducible --
backward.  SHA-256 of colour, accumulation, radii, n_contrib, final_T and every gradient are compared.
"""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child():
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "wild-gaussians_amd"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import wg_scenes as S
    from diff_gaussian_rasterization import _C
    from wg_testlib import run_hip, run_hip_native

    _C.set_option("deterministic_backward", 1)

    def sha(a):
        return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]

    scenes = []
    for name, (P, W, H, deg, sm) in {"config1_sh0_256": (10000, 256, 256, 0, 1.0), "sh3_640x360": (30000, 640, 360, 3, 1.5),
                                       "precomp_ragged_250x130": (8000, 250, 130, None, 3.0), "sh1_big_splats": (1500, 320, 200, 1, 12.0),
                                       "needles_500x300": (6000, 500, 300, None, 1.0), "bench_1080p": (1000000, 1920, 1080, 3, 1.0),
                                       "bench_1080p_x3": (1000000, 1920, 1080, None, 3.0), "wide_2560x1440": (400000, 2560, 1440, None, 2.0)}.items():
        cloud = S.make_cloud(P, W, H, sh_degree=deg, seed=11, scale_mult=sm)
        if name.startswith("needles"):
            cloud["scales"][:, 0] *= 60.0
            cloud["scales"][:, 1:] *= 0.5
        scenes.append((name, cloud, S.make_camera(W, H), deg if deg is not None else 0))
    out = {}
    for name, cloud, cam, deg in scenes:
        W, H = cam["width"], cam["height"]
        rng = np.random.default_rng(3)
        so = rng.uniform(-0.5, 0.5, size=(H, W, 2)).astype(np.float32) if ("ragged" in name or "needles" in name) else None
        bg = np.array([0.2, 0.5, 0.8], np.float32)
        h = run_hip(cloud, cam, sh_degree=deg, bg=bg, subpixel_offset=so, cotangent=S.make_cotangent(W, H))
        n = run_hip_native(cloud, cam, sh_degree=deg, bg=bg, subpixel_offset=so)
        im = n["views"]["image"]
        r = dict(color=sha(h["color"]), accumulation=sha(h["accumulation"]), radii=sha(h["radii"]), num_rendered=int(n["num_rendered"]),
                 n_contrib=sha(im["n_contrib"].cpu().numpy()), final_T=sha(im["final_T"].cpu().numpy()))
        for k, g in h["grads"].items():
            r["grad_" + k] = sha(g)
        out[name] = r
    print("AB_RESULT " + json.dumps(out))


def main():
    if len(sys.argv) == 2 and sys.argv[1] == "--child":
        return child()
    libs = sys.argv[1:3]
    res = []
    for lib in libs:
        env = dict(os.environ, WG_RASTERIZER_LIB=os.path.abspath(lib))
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True)
        line = [l for l in p.stdout.splitlines() if l.startswith("AB_RESULT ")]
        if p.returncode != 0 or not line:
            print(p.stdout[-2000:], p.stderr[-4000:])
            raise SystemExit(f"{lib}: the child failed")
        res.append(json.loads(line[0][len("AB_RESULT "):]))
    bad = 0
    for scene in res[0]:
        diff = [k for k in res[0][scene] if res[0][scene][k] != res[1][scene].get(k)]
        print(f"{scene:28s} {'IDENTICAL (' + str(len(res[0][scene])) + ' arrays)' if not diff else 'DIFFERENT: ' + ', '.join(diff)}")
        bad += len(diff)
    print(f"A = {libs[0]}\nB = {libs[1]}\n{'all outputs bit-identical' if bad == 0 else str(bad) + ' arrays differ'}")
    raise SystemExit(1 if bad else 0)


if __name__ == "__main__":
    main()
