#!/bin/bash
# round 4, GPU call 1: decision-exact compositing -- suite, A/B cost, sweep vs the reference's kernels
O=gpurun_out/r4c1; mkdir -p $O
python -m pytest tests -m gpu -q -x --timeout 900 > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" | tee -a $O/summary.txt; tail -15 $O/pytest_gpu.log
B="--steps 300 --warmup 30 --no-cpu-baseline"
for rep in 1 2; do
  python bench.py $B > $O/exact.$rep.json 2> $O/exact.$rep.err
  python bench.py $B --option exact_compositing=0 > $O/fast.$rep.json 2> $O/fast.$rep.err
  WG_RASTERIZER_LIB=$PWD/wild-gaussians_amd/build/bwd6/libwg_rasterizer.so python bench.py $B > $O/exact_bwd6.$rep.json 2> $O/exact_bwd6.$rep.err
done
python - $O <<'PY' | tee -a $O/summary.txt
import json,sys,glob
for f in sorted(glob.glob(sys.argv[1]+"/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); s=d.get("stages_ms",{})
        print(f"{f.split('/')[-1]:22s} train {d['value']:8.1f} it/s fwd {d.get('forward_fps',0):8.1f} fps render_fwd {s.get('render_forward',0):.4f} render_bwd {s.get('render_backward',0):.4f}", d.get("reference_on_this_gpu",{}).get("product_vs_reference"))
    except Exception as e: print(f, "FAILED", e)
PY
python tests/tools/stress_sweep_vs_reference.py 100000 3000 > $O/sweep_3k.txt 2>&1; tail -3 $O/sweep_3k.txt
