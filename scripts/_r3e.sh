mkdir -p gpurun_out/r3e
(time python -m pytest tests -m gpu -q --durations=5) > gpurun_out/r3e/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3e/pytest.log
tail -12 gpurun_out/r3e/pytest.log
for rep in 1 2; do
python bench.py --steps 300 --warmup 50 --no-cpu-baseline > gpurun_out/r3e/headline.$rep.json 2>/dev/null
python bench.py --steps 300 --warmup 50 --no-cpu-baseline --option deterministic_backward=1 > gpurun_out/r3e/headline_det.$rep.json 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3e/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); s=d.get('stages_ms',{})
        print(f.split('/')[-1], 'value',d['value'],'fwd',d.get('forward_fps'),'ms',d['ms_per_step'], s)
    except Exception as e: print(f,'FAIL',e)
PY
