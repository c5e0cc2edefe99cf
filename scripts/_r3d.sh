mkdir -p gpurun_out/r3d
(time python -m pytest tests/test_parity_gpu.py tests/test_real_caller.py -m gpu -q --durations=5 -x) > gpurun_out/r3d/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3d/pytest.log
tail -12 gpurun_out/r3d/pytest.log
for cfg in "geometry_reuse=0 speculative_forward=0" "geometry_reuse=0 speculative_forward=1" "geometry_reuse=1 speculative_forward=1"; do
  opts=""; for o in $cfg; do opts="$opts --option $o"; done
  tag=$(echo $cfg | tr ' =' '__')
  python scripts/bench_wildgaussians_step.py --real-caller --optins --steps 20 --warmup 5 $opts > gpurun_out/r3d/real_$tag.json 2> gpurun_out/r3d/real_$tag.err
  python scripts/bench_wildgaussians_step.py --real-caller --steps 10 --warmup 3 $opts > gpurun_out/r3d/realplain_$tag.json 2> gpurun_out/r3d/realplain_$tag.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3d/real*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], d['train_step_ms'], d['rasterizer_only_ms (2 fwd + 2 bwd, incl. input clones)'], d['library'])
    except Exception as e: print(f,'FAIL',e)
PY
