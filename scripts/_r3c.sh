mkdir -p gpurun_out/r3c
(time python -m pytest tests -m gpu -q --durations=5) > gpurun_out/r3c/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3c/pytest.log
for rep in 1 2; do for sp in 1 0; do
python bench.py --steps 300 --warmup 50 --no-cpu-baseline --option speculative_forward=$sp > gpurun_out/r3c/headline_spec$sp.$rep.json 2>/dev/null
python bench.py --steps 200 --warmup 30 --no-cpu-baseline --scale-mult 3 --option speculative_forward=$sp > gpurun_out/r3c/x3_spec$sp.$rep.json 2>/dev/null
python bench.py --steps 50 --warmup 10 --no-cpu-baseline --gaussians 10000000 --width 3840 --height 2160 --forward-only --option speculative_forward=$sp > gpurun_out/r3c/c5_spec$sp.$rep.json 2>/dev/null
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3c/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); s=d.get('stages_ms',{})
        print(f.split('/')[-1], 'value',d['value'],'fwd',d.get('forward_fps'),'ms',d['ms_per_step'],'fwd_ms',d.get('forward_ms'), 'q',d.get('step_ms_quantiles'))
    except Exception as e: print(f,'FAIL',e)
PY
tail -15 gpurun_out/r3c/pytest.log
