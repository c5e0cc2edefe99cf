O=gpurun_out/r6e; mkdir -p $O
wild-gaussians_amd/build/nan_min_probe 2>&1 | tee $O/nan_min_probe.txt
(time python bench.py --steps 20 --warmup 5) > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; tail -3 $O/bench_driver_cmd.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6e/bench_driver_cmd.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d.get('roofline',{}).get('frac'))
print(json.dumps(d.get('configs'), indent=1))
PY
