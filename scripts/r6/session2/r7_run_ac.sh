# what would a front sort with KNOWN thresholds be worth at the headline?  The lazy path's stage times there (its selection cost is what a per-camera depth history would remove)
O=gpurun_out/r7ac; mkdir -p $O
for rep in 1 2; do
for v in "" "--option lazy_min_len=512 --option lazy_target=400" "--option lazy_min_len=512 --option lazy_target=300" "--option lazy_min_len=256 --option lazy_target=300"; do
  python bench.py --no-cpu-baseline --no-camera-sequence --no-config-legs --steps 300 --warmup 50 $v > $O/ab.json 2>$O/ab.err || tail -3 $O/ab.err
  python - "$v" <<'PY'
import json,sys
d=json.loads(open('gpurun_out/r7ac/ab.json').read().strip().splitlines()[-1]); s=d.get('stages_ms',{})
print(f"{sys.argv[1]:56s} train {d['value']:8.1f} fwd {d['forward_fps']:8.1f} scan {s.get('scan',0):.4f} scat {s.get('duplicate_keys',0):.4f} sort {s.get('sort',0):.4f} K8 {s.get('render_forward',0):.4f} fix {s.get('render_fixup',0):.4f} K9 {s.get('render_backward',0):.4f}")
PY
done; done | tee $O/summary.txt
