#!/bin/bash
mkdir -p gpurun_out/r3h
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_reference_golden.py -q -m gpu -x 2>&1 | tail -4
AB_IDENTICAL=1 scripts/ab_run.sh gpurun_out/r3h "--steps 300 --warmup 50" scan0
for v in default scan0; do
  LIB=wild-gaussians_amd/build/$v/libwg_rasterizer.so; [ $v = default ] && LIB=wild-gaussians_amd/diff_gaussian_rasterization/libwg_rasterizer.so
  for cfg in "c5 --gaussians 10000000 --width 3840 --height 2160 --forward-only --steps 60 --warmup 10" "x3 --scale-mult 3 --colors precomp --steps 200 --warmup 30"; do
    set -- $cfg; name=$1; shift
    WG_RASTERIZER_LIB=$PWD/$LIB python bench.py --no-cpu-baseline "$@" > gpurun_out/r3h/${name}_$v.json 2> gpurun_out/r3h/${name}_$v.err
    python - gpurun_out/r3h/${name}_$v.json $name $v <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); s=d['stages_ms']
    print(sys.argv[2], sys.argv[3], d['value'], d.get('forward_fps'), 'scan', s['scan'], 'scatter', s['duplicate_keys'], 'sort', s['sort'])
except Exception as e: print(sys.argv[2], sys.argv[3], "FAILED", e)
PY
  done
done
