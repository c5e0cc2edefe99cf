mkdir -p gpurun_out/r3i
(time python -m pytest tests -m gpu -q -x) > gpurun_out/r3i/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3i/pytest.log
tail -6 gpurun_out/r3i/pytest.log
for rep in 1 2 3; do
python bench.py --steps 300 --warmup 50 --no-cpu-baseline > gpurun_out/r3i/headline.$rep.json 2>/dev/null
python bench.py --steps 300 --warmup 50 --no-cpu-baseline --option fused_scan=1 > gpurun_out/r3i/headline_fused.$rep.json 2>/dev/null
done
python bench.py --steps 50 --warmup 10 --no-cpu-baseline --gaussians 10000000 --width 3840 --height 2160 --forward-only > gpurun_out/r3i/c5.json 2>/dev/null
python bench.py --steps 50 --warmup 10 --no-cpu-baseline --gaussians 10000000 --width 3840 --height 2160 --forward-only --option fused_scan=1 > gpurun_out/r3i/c5_fused.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3i/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); s=d.get('stages_ms',{})
        print(f.split('/')[-1], 'value',d['value'],'fwd',d.get('forward_fps'),'ms',d['ms_per_step'], 'bwd', s.get('render_backward'), 'scan', s.get('scan'), 'order', s.get('tile_ranges'))
    except Exception as e: print(f,'FAIL',e)
PY
