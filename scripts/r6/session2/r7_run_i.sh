# round 6, second session: sh_stream option -- the new tests, where the 1 M gain comes from, the driver's line
O=gpurun_out/r7i; mkdir -p $O
python -m pytest tests/test_parity_gpu.py -q -m gpu -x -k "non_temporal or near_aim or argument_sweep" 2>&1 | tail -3
python scripts/r6/diag_sh_stream.py > $O/diag_sh_stream_1M.json 2> $O/diag.err; cat $O/diag_sh_stream_1M.json | tr -d '\n' | sed 's/},/},\n/g'; echo
python bench.py --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
python - $O/bench_driver_cmd.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['stages_ms'], {k:(v.get('iters_per_s') or v.get('fps'), v.get('pins_ok')) for k,v in d['configs'].items() if isinstance(v,dict)})
PY
