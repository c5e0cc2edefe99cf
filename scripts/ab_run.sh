#!/bin/bash
# On the GPU box: bench every variant library under wild-gaussians_amd/build/<name>/ named on the command line, plus the in-tree
# one ("default"), and print one line each (train iter/s, forward fps, render stage times); optionally check bit-identity.
# usage: scripts/ab_run.sh <outdir> "<bench args>" name1 name2 ...     (AB_IDENTICAL=1: tests/tools/ab_bit_identical.py vs default)
OUT=$1; ARGS=$2; shift 2
mkdir -p $OUT
for rep in 1 2; do
for v in default "$@"; do
  LIB=wild-gaussians_amd/build/$v/libwg_rasterizer.so
  [ $v = default ] && LIB=wild-gaussians_amd/diff_gaussian_rasterization/libwg_rasterizer.so
  WG_RASTERIZER_LIB=$PWD/$LIB python bench.py --no-cpu-baseline $ARGS > $OUT/$v.$rep.json 2> $OUT/$v.$rep.err
  python - $OUT/$v.$rep.json $v <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); s=d.get("stages_ms",{})
    print(f"{sys.argv[2]:24s} train {d['value']:8.1f} it/s  fwd {d.get('forward_fps',0):8.1f} fps  render_fwd {s.get('render_forward',0):.4f}  render_bwd {s.get('render_backward',0):.4f}  sort {s.get('sort',0):.4f} scan {s.get('scan',0):.4f} scatter {s.get('duplicate_keys',0):.4f} pre {s.get('preprocess',0):.4f} prebwd {s.get('preprocess_backward',0):.4f}")
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done; done | tee -a $OUT/summary.txt
if [ -n "$AB_IDENTICAL" ]; then
  for v in "$@"; do
    python tests/tools/ab_bit_identical.py wild-gaussians_amd/diff_gaussian_rasterization/libwg_rasterizer.so wild-gaussians_amd/build/$v/libwg_rasterizer.so 2>&1 | tail -4 | tee -a $OUT/summary.txt
  done
fi
