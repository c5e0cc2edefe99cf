# round 6, second session: option sweeps at the headline (no code change): forced near / far split, band lists at 1 M Gaussians
O=gpurun_out/r7z; mkdir -p $O
for rep in 1 2; do
for v in "" "--option band_list_min_p=1" "--option near_split=1" "--option near_split=1 --option band_list_min_p=1" "--option near_split=1 --option near_per_tile=350" "--option near_split=1 --option near_per_tile=500"; do
  python bench.py --no-cpu-baseline --no-camera-sequence --no-config-legs --steps 300 --warmup 50 $v > $O/ab.json 2>$O/ab.err || tail -3 $O/ab.err
  python - "$v" <<'PY'
import json,sys
d=json.loads(open('gpurun_out/r7z/ab.json').read().strip().splitlines()[-1]); s=d.get('stages_ms',{})
print(f"{sys.argv[1]:60s} train {d['value']:8.1f} fwd {d['forward_fps']:8.1f} pre {s.get('preprocess',0):.4f} scan {s.get('scan',0):.4f} scat {s.get('duplicate_keys',0):.4f} sort {s.get('sort',0):.4f} K8 {s.get('render_forward',0):.4f} fix {s.get('render_fixup',0):.4f} K9 {s.get('render_backward',0):.4f}")
PY
done; done | tee $O/summary.txt
