mkdir -p gpurun_out/r3k
python bench.py > gpurun_out/r3k/bench_final.json 2> gpurun_out/r3k/bench_final.err
python bench.py --gaussians 10000000 --width 3840 --height 2160 --forward-only --steps 100 --warmup 20 > gpurun_out/r3k/bench_config5.json 2> gpurun_out/r3k/bench_config5.err
python bench.py --gaussians 500000 --steps 300 --warmup 50 > gpurun_out/r3k/bench_config2.json 2>/dev/null
python bench.py --gaussians 3000000 --width 1600 --height 1200 --colors precomp --steps 200 --warmup 30 --no-cpu-baseline > gpurun_out/r3k/bench_3M.json 2>/dev/null
python bench.py --gaussians 3000000 --width 1600 --height 1200 --colors precomp --scale-mult 2 --steps 200 --warmup 30 --no-cpu-baseline > gpurun_out/r3k/bench_3M_x2.json 2>/dev/null
for m in 2 3 4; do python bench.py --scale-mult $m --steps 200 --warmup 30 --no-cpu-baseline > gpurun_out/r3k/bench_dense_x$m.json 2>/dev/null; done
WG_DIST_BACKEND=gloo python bench.py --gpus 2 --steps 100 --warmup 20 --baseline-iters-per-s 975 > gpurun_out/r3k/bench_gpus2_one_device_gloo.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3k/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); s=d.get('stages_ms',{})
        print(f.split('/')[-1], 'value',d['value'],'fwd',d.get('forward_fps'),'ms',d['ms_per_step'], 'roofline', {k:d.get('roofline',{}).get(k) for k in ('bound','kernel','frac')}, d.get('speedup_vs_reference_on_this_gpu'), d.get('scaling_efficiency'))
    except Exception as e: print(f,'FAIL',e)
PY
