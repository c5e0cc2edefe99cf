mkdir -p gpurun_out/r6b; O=gpurun_out/r6b
(time python -m pytest tests -m gpu -x -q) > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
python tests/tools/nonfinite_inputs.py > $O/nonfinite.log 2>&1; tail -25 $O/nonfinite.log
python scripts/probe_balance.py --dump $O/probe_order_on.npz > $O/probe_order_on.json 2>$O/probe.err; cat $O/probe_order_on.json
python scripts/probe_balance.py --option forward_order=0 > $O/probe_order_off.json 2>>$O/probe.err; cat $O/probe_order_off.json
python scripts/probe_balance.py --option backward_order_period=128 > $O/probe_bwd_snake.json 2>>$O/probe.err; cat $O/probe_bwd_snake.json
python scripts/probe_balance.py --option order_period=0 > $O/probe_order_desc.json 2>>$O/probe.err; cat $O/probe_order_desc.json
for rep in 1 2; do
for v in "" "--option forward_order=0" "--option backward_order_period=128"; do
  python bench.py --no-cpu-baseline --no-camera-sequence --steps 300 --warmup 50 $v > $O/ab.json 2>$O/ab.err
  python - "$v" <<'PY'
import json,sys
d=json.loads(open('gpurun_out/r6b/ab.json').read().strip().splitlines()[-1]); s=d.get('stages_ms',{})
print(f"{sys.argv[1]:40s} train {d['value']:8.1f} fwd {d['forward_fps']:8.1f} K8 {s.get('render_forward',0):.4f} K9 {s.get('render_backward',0):.4f} ranges {s.get('tile_ranges',0):.4f} pre {s.get('preprocess',0):.4f}")
PY
done; done | tee $O/ab_summary.txt
