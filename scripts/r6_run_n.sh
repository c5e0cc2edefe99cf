O=gpurun_out/r6n; mkdir -p $O
python -m pytest tests/test_parity_gpu.py -q -m gpu -x -k "launch_order or config2 or full_size" 2>&1 | tail -5
for rep in 1 2; do
for v in "" "--option band_balance=0"; do
  python bench.py --no-cpu-baseline --no-camera-sequence --no-config-legs --steps 300 --warmup 50 $v > $O/ab.json 2>$O/ab.err || tail -3 $O/ab.err
  python - "$v" <<'PY'
import json,sys
d=json.loads(open('gpurun_out/r6n/ab.json').read().strip().splitlines()[-1]); s=d.get('stages_ms',{})
print(f"{sys.argv[1]:34s} train {d['value']:8.1f} fwd {d['forward_fps']:8.1f} K8 {s.get('render_forward',0):.4f} K9 {s.get('render_backward',0):.4f} ranges {s.get('tile_ranges',0):.4f} scan {s.get('scan',0):.4f}")
PY
done; done | tee $O/summary.txt
python scripts/probe_balance.py > $O/probe_balance.json 2>$O/probe.err; python scripts/probe_balance.py --option band_balance=0 > $O/probe_nobalance.json 2>>$O/probe.err
python - <<'PY'
import json
for f in ('probe_balance','probe_nobalance'):
    d=json.loads(open(f'gpurun_out/r6n/{f}.json').read())
    print(f, {k:(d[k]['kernel_span_us'], d[k]['mean_over_max_simd_finish']) for k in ('render_forward','render_backward')})
PY
