#!/bin/bash
# round 4, GPU call 3: issue priority for long walks (s_setprio) and parking polled every batch
O=gpurun_out/r4c3; mkdir -p $O
python - > $O/tile_last_quantiles.txt 2>&1 <<'PY'
import sys, os
sys.path.insert(0, "wild-gaussians_amd"); sys.path.insert(0, ".")
import numpy as np, torch
import wg_scenes as S
from tests.wg_testlib import run_hip_native
W, H, P = 1920, 1080, 1_000_000
n = run_hip_native(S.make_cloud(P, W, H, sh_degree=3, seed=0), S.make_camera(W, H), sh_degree=3)
tl = n["views"]["image"]["tile_last"].cpu().numpy().astype(np.int64)
rg = n["views"]["image"]["ranges"].cpu().numpy().astype(np.int64); ln = rg[:, 1] - rg[:, 0]
q = [0.1, 0.25, 0.5, 0.75, 0.9, 0.95, 0.99, 1.0]
print("tile_last quantiles", dict(zip(q, np.quantile(tl, q).tolist())), "mean", tl.mean())
print("list length quantiles", dict(zip(q, np.quantile(ln, q).tolist())), "mean", ln.mean())
print("corr(tile_last, len)", np.corrcoef(tl, ln)[0, 1])
PY
cat $O/tile_last_quantiles.txt
B="--steps 300 --warmup 30 --no-cpu-baseline"
run() { # name lib opts
  LIB=wild-gaussians_amd/build/$2/libwg_rasterizer.so; [ $2 = default ] && LIB=wild-gaussians_amd/diff_gaussian_rasterization/libwg_rasterizer.so
  WG_RASTERIZER_LIB=$PWD/$LIB timeout 300 python bench.py $B $3 > $O/$1.json 2> $O/$1.err
}
for rep in 1 2; do
  run default.$rep default ""
  for v in prio128 prio256 prio384 prio512; do run $v.$rep $v ""; done
  run park85.$rep default "--option forward_parking=85"
  run park92.$rep default "--option forward_parking=92"
  run prio256_park92.$rep prio256 "--option forward_parking=92"
done
python - $O <<'PY' | tee -a $O/summary.txt
import json,sys,glob
for f in sorted(glob.glob(sys.argv[1]+"/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); s=d.get("stages_ms",{})
        print(f"{f.split('/')[-1]:24s} train {d['value']:8.1f} it/s fwd {d.get('forward_fps',0):8.1f} fps render_fwd {s.get('render_forward',0):.4f} render_bwd {s.get('render_backward',0):.4f}")
    except Exception as e: print(f, "FAILED", e)
PY
