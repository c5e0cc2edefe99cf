O=gpurun_out/r6k; mkdir -p $O
for rep in 1 2 3; do
  python bench.py --no-cpu-baseline --no-camera-sequence --no-config-legs --steps 300 --warmup 50 > $O/ab.json 2>$O/ab.err
  python - <<'PY'
import json,sys
d=json.loads(open('gpurun_out/r6k/ab.json').read().strip().splitlines()[-1]); s=d.get('stages_ms',{})
print(f"train {d['value']:8.1f} fwd {d['forward_fps']:8.1f} K8 {s.get('render_forward',0):.4f} K9 {s.get('render_backward',0):.4f} ranges {s.get('tile_ranges',0):.4f} scan {s.get('scan',0):.4f} pre {s.get('preprocess',0):.4f} prebwd {s.get('preprocess_backward',0):.4f}")
PY
done | tee $O/summary.txt
(time python -m pytest tests -m gpu -x -q) > $O/pytest_gpu.log 2>&1; grep -n "passed\|failed" $O/pytest_gpu.log | tail -2
