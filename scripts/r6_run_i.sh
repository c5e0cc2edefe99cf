O=gpurun_out/r6i; mkdir -p $O
for v in "" "--option near_per_tile=500" "--option near_per_tile=450"; do
  python bench.py --gaussians 10000000 --width 3840 --height 2160 --forward-only --steps 100 --warmup 20 --no-cpu-baseline $v > $O/c5.json 2>$O/c5.err || tail -3 $O/c5.err
  python - "$v" <<'PY'
import json,sys
d=json.loads(open('gpurun_out/r6i/c5.json').read().strip().splitlines()[-1]); s=d.get('stages_ms',{})
print(f"{sys.argv[1]:40s} fps {d['value']:7.1f} ms {d['ms_per_step']:.4f} pre {s.get('preprocess',0):.3f} scan {s.get('scan',0):.3f} scatter {s.get('duplicate_keys',0):.3f} sort {s.get('sort',0):.3f} render {s.get('render_forward',0):.3f} fixup {s.get('render_fixup',0):.3f}", d['library']['near_far_split'])
PY
done | tee $O/c5_summary.txt
