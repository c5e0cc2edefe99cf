# round 6, second session: from how many Gaussians on do the band lists pay?  (option sweep, no code change)
O=gpurun_out/r7aa; mkdir -p $O
for cfg in "--gaussians 2000000" "--gaussians 3000000 --width 1600 --height 1200 --colors precomp" "--gaussians 3000000" "--gaussians 5000000"; do
echo "== $cfg"
for rep in 1 2; do
for v in "" "--option band_list_min_p=4000000" "--option band_list_min_p=8000000"; do
  python bench.py $cfg --no-cpu-baseline --no-camera-sequence --no-config-legs --steps 150 --warmup 40 $v > $O/ab.json 2>$O/ab.err || tail -3 $O/ab.err
  python - "$v" <<'PY'
import json,sys
d=json.loads(open('gpurun_out/r7aa/ab.json').read().strip().splitlines()[-1]); s=d.get('stages_ms',{})
print(f"{sys.argv[1]:40s} train {d['value']:8.1f} fwd {d['forward_fps']:8.1f} pre {s.get('preprocess',0):.4f} scan {s.get('scan',0):.4f} scat {s.get('duplicate_keys',0):.4f} sort {s.get('sort',0):.4f} K8 {s.get('render_forward',0):.4f} fix {s.get('render_fixup',0):.4f} K9 {s.get('render_backward',0):.4f}")
PY
done; done; done | tee $O/summary.txt
