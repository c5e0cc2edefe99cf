mkdir -p gpurun_out/r3l
python bench.py --gaussians 3000000 --width 1600 --height 1200 --colors precomp --steps 200 --warmup 30 --no-cpu-baseline > gpurun_out/r3l/bench_3M.json 2>/dev/null
python bench.py --gaussians 3000000 --width 1600 --height 1200 --colors precomp --scale-mult 2 --steps 200 --warmup 30 --no-cpu-baseline > gpurun_out/r3l/bench_3M_x2.json 2>/dev/null
python bench.py --steps 100 --warmup 20 --no-cpu-baseline > gpurun_out/r3l/bench_headline_check.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3l/*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['value'], d.get('forward_fps'), d['stages_ms'], d['library'].get('geometry_reuse'))
PY
