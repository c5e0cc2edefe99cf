# round 6, second session: near-aim controller before / after "any asking tile is a failure", 7000 pipelined frames each; GPU suite on the generalised K9 reduction
O=gpurun_out/r7f; mkdir -p $O
WG_RASTERIZER_LIB=$PWD/wild-gaussians_amd/build/policy_before/libwg_rasterizer.so NEAR_TRACE_BENCH_LIKE=1 python scripts/r6/near_trace.py 3500 > $O/near_trace_before.txt 2> $O/near_trace_before.err; head -1 $O/near_trace_before.txt
NEAR_TRACE_BENCH_LIKE=1 python scripts/r6/near_trace.py 3500 > $O/near_trace_after.txt 2> $O/near_trace_after.err; head -1 $O/near_trace_after.txt
(time python -m pytest tests -m gpu -q -x) > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
python bench.py --gaussians 10000000 --width 3840 --height 2160 --forward-only --no-cpu-baseline --no-camera-sequence --steps 1500 --warmup 20 > $O/c5_1500.json 2> $O/c5_1500.err
python - $O/c5_1500.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], d['value'], d['forward_fps'], d['step_ms_quantiles'], d['forward_ms_quantiles'], d['library']['near_far_split'])
PY
python bench.py --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
python - $O/bench_driver_cmd.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['stages_ms'])
PY
for a in "--views 8 --streams 3" "--option deterministic_backward=1"; do
python bench.py --no-cpu-baseline --no-camera-sequence --no-config-legs --steps 200 --warmup 30 $a > $O/b.json 2> $O/b.err
python - $O/b.json "$a" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], d['value'], d['ms_per_step'], d['stages_ms'])
PY
done
python scripts/bench_two_tone_call.py > $O/two_tone.json 2> $O/two_tone.err; tail -c 1500 $O/two_tone.json
