# parity of a training step at 5 M Gaussians (lazy colour + band lists + split active) against the CPU oracle and the reference build
O=gpurun_out/r7ab; mkdir -p $O
python bench.py --gaussians 5000000 --steps 50 --warmup 10 --no-camera-sequence > $O/bench_5M.json 2> $O/bench_5M.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r7ab/bench_5M.json').read().strip().splitlines()[-1])
print(d['value'], d['forward_fps'], d['stages_ms'])
print(json.dumps(d.get('parity'))[:900])
r=d.get('reference_on_this_gpu',{}); print(json.dumps(r.get('product_vs_reference'))[:700])
PY
