O=gpurun_out/r7l; mkdir -p $O
python bench.py --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
python - $O/bench_driver_cmd.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['stages_ms'])
for k,v in d['configs'].items():
    if isinstance(v,dict): print(k, {kk:v.get(kk) for kk in ('iters_per_s','fps','ms_per_step','steps','pins_ok','stages_ms','near_far_split')})
PY
bash scripts/ab_run.sh $O "--no-camera-sequence --no-config-legs --steps 300 --warmup 50" ntout
