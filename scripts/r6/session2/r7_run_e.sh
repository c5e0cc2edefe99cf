# round 6, second session: full GPU suite on the K9 LDS reduction + the controller whose reports carry their frame's aim; config 5 over long runs
O=gpurun_out/r7e; mkdir -p $O
(time python -m pytest tests -m gpu -q -x) > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
NEAR_TRACE_BENCH_LIKE=1 python scripts/r6/near_trace.py 1500 > $O/near_trace_bench_like.txt 2> $O/near_trace.err; head -1 $O/near_trace_bench_like.txt; sed -n 2,40p $O/near_trace_bench_like.txt | cut -c1-120
for K in 100 1500; do
python bench.py --gaussians 10000000 --width 3840 --height 2160 --forward-only --no-cpu-baseline --no-camera-sequence --steps $K --warmup 20 > $O/c5_$K.json 2> $O/c5_$K.err
python - $O/c5_$K.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], d['value'], d['forward_fps'], d['step_ms_quantiles'], d['library']['near_far_split'], d['stages_ms'])
PY
done
python bench.py --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
python - $O/bench_driver_cmd.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline'], {k:(v.get('iters_per_s') or v.get('fps'), v.get('pins_ok')) for k,v in d.get('configs',{}).items()})
PY
