#!/bin/bash
# round 4, GPU call 10: occupancy of the two-colour kernels (A/B), ticket / stats fixes through the suite's deferred + fixed tests
O=gpurun_out/r4c10; mkdir -p $O
python -m pytest tests/test_parity_gpu.py -m gpu -q -x --timeout 900 -k "deferred or fixed or speculative or two_colour or reuse" > $O/pytest_sel.log 2>&1; echo "pytest rc $?" | tee $O/summary.txt; tail -3 $O/pytest_sel.log
for rep in 1 2; do
for v in default dualf6 dualb5; do
  LIB=wild-gaussians_amd/build/$v/libwg_rasterizer.so; [ $v = default ] && LIB=wild-gaussians_amd/diff_gaussian_rasterization/libwg_rasterizer.so
  WG_RASTERIZER_LIB=$PWD/$LIB timeout 600 python scripts/bench_wildgaussians_step.py --real-caller --steps 30 --warmup 2 --dual > $O/$v.$rep.json 2> $O/$v.$rep.err
  tail -1 $O/$v.$rep.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', {k:v for k,v in d.items() if 'ms' in k})" | tee -a $O/summary.txt
done; done
