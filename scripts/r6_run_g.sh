O=gpurun_out/r6g; mkdir -p $O
for v in "" "--option near_per_tile=700" "--option near_per_tile=500" "--option near_per_tile=400" "--option near_per_tile=300" "--option near_per_tile=500 --option lazy_target=460" "--option near_per_tile=400 --option lazy_target=380" "--option forward_order=0"; do
  python bench.py --gaussians 10000000 --width 3840 --height 2160 --forward-only --steps 60 --warmup 15 --no-cpu-baseline $v > $O/c5.json 2>$O/c5.err || tail -3 $O/c5.err
  python - "$v" <<'PY'
import json,sys
d=json.loads(open('gpurun_out/r6g/c5.json').read().strip().splitlines()[-1]); s=d.get('stages_ms',{})
print(f"{sys.argv[1]:55s} fps {d['value']:7.1f} ms {d['ms_per_step']:.4f} pre {s.get('preprocess',0):.3f} scan {s.get('scan',0):.3f} scatter {s.get('duplicate_keys',0):.3f} sort {s.get('sort',0):.3f} render {s.get('render_forward',0):.3f} fixup {s.get('render_fixup',0):.3f}")
PY
done | tee $O/c5_summary.txt
