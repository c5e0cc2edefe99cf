O=gpurun_out/r6m; mkdir -p $O
python -m pytest tests/test_parity_gpu.py -q -m gpu -k "side_stream" 2>&1 | tail -2
for v in "--option split_preprocess=0" "--option split_waves=256" "--option split_waves=512" "--option split_waves=1024" "--option split_waves=2048" "--option split_waves=4096"; do
  python bench.py --no-cpu-baseline --no-camera-sequence --no-config-legs --steps 300 --warmup 50 $v > $O/ab.json 2>$O/ab.err || tail -3 $O/ab.err
  python - "$v" <<'PY'
import json,sys
d=json.loads(open('gpurun_out/r6m/ab.json').read().strip().splitlines()[-1]); s=d.get('stages_ms',{})
print(f"{sys.argv[1]:34s} train {d['value']:8.1f} fwd {d['forward_fps']:8.1f} K8 {s.get('render_forward',0):.4f} K9 {s.get('render_backward',0):.4f} scan {s.get('scan',0):.4f} scatter {s.get('duplicate_keys',0):.4f} sort {s.get('sort',0):.4f} pre {s.get('preprocess',0):.4f}")
PY
done | tee $O/summary.txt
for v in "--option split_preprocess=0" "--option split_waves=512" "--option split_waves=1024" "--option split_waves=2048"; do
  python bench.py --gaussians 10000000 --width 3840 --height 2160 --forward-only --steps 100 --warmup 20 --no-cpu-baseline $v > $O/c5.json 2>$O/c5.err || tail -3 $O/c5.err
  python - "$v" <<'PY'
import json,sys
d=json.loads(open('gpurun_out/r6m/c5.json').read().strip().splitlines()[-1]); s=d.get('stages_ms',{})
print(f"{sys.argv[1]:34s} fps {d['value']:7.1f} ms {d['ms_per_step']:.4f} pre {s.get('preprocess',0):.3f} scan {s.get('scan',0):.3f} scatter {s.get('duplicate_keys',0):.3f} sort {s.get('sort',0):.3f} render {s.get('render_forward',0):.3f}")
PY
done | tee -a $O/summary.txt
