mkdir -p gpurun_out/r6c; O=gpurun_out/r6c
(time python -m pytest tests -m gpu -x -q) > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
python tests/tools/nonfinite_inputs.py 2>&1 | grep -v "libdrm" > $O/nonfinite.log; grep -c "over 1e-4: 0 " $O/nonfinite.log; grep "rot NaN\|opacity NaN" $O/nonfinite.log
for rep in 1 2; do
for v in "" "--option forward_order=0" "--option order_period=0" "--option order_period=64" "--option order_period=256"; do
  python bench.py --no-cpu-baseline --no-camera-sequence --steps 300 --warmup 50 $v > $O/ab.json 2>$O/ab.err
  python - "$v" <<'PY'
import json,sys
d=json.loads(open('gpurun_out/r6c/ab.json').read().strip().splitlines()[-1]); s=d.get('stages_ms',{})
print(f"{sys.argv[1]:40s} train {d['value']:8.1f} fwd {d['forward_fps']:8.1f} K8 {s.get('render_forward',0):.4f} K9 {s.get('render_backward',0):.4f} ranges {s.get('tile_ranges',0):.4f} scan {s.get('scan',0):.4f} pre {s.get('preprocess',0):.4f}")
PY
done; done | tee $O/ab_summary.txt
python bench.py --views 8 --gpus 1 --no-cpu-baseline > $O/views8.json 2>$O/views8.err; python -c "
import json; d=json.loads(open('$O/views8.json').read().strip().splitlines()[-1]); print('views8', d['value'], d['ms_per_step'])"
python bench.py --views 8 --gpus 1 --no-cpu-baseline --option forward_order=0 > $O/views8_off.json 2>$O/views8.err; python -c "
import json; d=json.loads(open('$O/views8_off.json').read().strip().splitlines()[-1]); print('views8 order off', d['value'], d['ms_per_step'])"
