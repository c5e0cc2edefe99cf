cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -x -q --tb=short -k "near_far or backs_off" 2>&1 | tail -15
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
B="python bench.py --no-cpu-baseline"
run() { n=$1; shift; timeout 600 $B "$@" > gpurun_out/bench_$n.log 2>&1; tail -1 gpurun_out/bench_$n.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n', d['value'], d['forward_fps'], d['stages_ms'])" || tail -5 gpurun_out/bench_$n.log; }
run r2c34_headline --steps 200 --warmup 30
run r2c34_x3 --steps 100 --warmup 10 --scale-mult 3
