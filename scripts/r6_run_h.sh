O=gpurun_out/r6h; mkdir -p $O
(time python -m pytest tests -m gpu -x -q) > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
for v in "" "--option near_adapt=0" "--option near_per_tile=500"; do
  python bench.py --gaussians 10000000 --width 3840 --height 2160 --forward-only --steps 100 --warmup 20 --no-cpu-baseline $v > $O/c5.json 2>$O/c5.err || tail -3 $O/c5.err
  python - "$v" <<'PY'
import json,sys
d=json.loads(open('gpurun_out/r6h/c5.json').read().strip().splitlines()[-1]); s=d.get('stages_ms',{})
print(f"{sys.argv[1]:40s} fps {d['value']:7.1f} ms {d['ms_per_step']:.4f} pre {s.get('preprocess',0):.3f} scan {s.get('scan',0):.3f} scatter {s.get('duplicate_keys',0):.3f} sort {s.get('sort',0):.3f} render {s.get('render_forward',0):.3f} fixup {s.get('render_fixup',0):.3f}", d['forward_ms_quantiles'])
PY
done | tee $O/c5_summary.txt
for v in "--scale-mult 3" "--scale-mult 3 --option near_adapt=0" "--gaussians 3000000 --width 1600 --height 1200 --colors precomp --scale-mult 2" "--gaussians 3000000 --width 1600 --height 1200 --colors precomp --scale-mult 2 --option near_adapt=0"; do
  python bench.py --steps 150 --warmup 30 --no-cpu-baseline --no-camera-sequence $v > $O/d.json 2>$O/d.err || tail -3 $O/d.err
  python - "$v" <<'PY'
import json,sys
d=json.loads(open('gpurun_out/r6h/d.json').read().strip().splitlines()[-1]); s=d.get('stages_ms',{})
print(f"{sys.argv[1]:100s} train {d['value']:7.1f} fwd {d['forward_fps']:7.1f} scan {s.get('scan',0):.3f} scatter {s.get('duplicate_keys',0):.3f} sort {s.get('sort',0):.3f} render {s.get('render_forward',0):.3f} fixup {s.get('render_fixup',0):.3f}")
PY
done | tee $O/dense_summary.txt
