cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2c11; mkdir -p $O
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -x -q --tb=short -k "binning_bit_exact or near_far or full_size" 2>&1 | tail -15
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
B="python bench.py --no-cpu-baseline"
run() { n=$1; shift; timeout 600 $B "$@" > $O/bench_$n.log 2>&1; tail -1 $O/bench_$n.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n', d['value'], d['forward_fps'], d['stages_ms'])" || tail -5 $O/bench_$n.log; }
run c5_box --steps 30 --warmup 5 --gaussians 10000000 --width 3840 --height 2160 --forward-only
run c5_nobox --steps 30 --warmup 5 --gaussians 10000000 --width 3840 --height 2160 --forward-only --option box_count=0
run x3_box --steps 100 --warmup 10 --scale-mult 3
run x3_nobox --steps 100 --warmup 10 --scale-mult 3 --option box_count=0
run headline_box --steps 200 --warmup 30 --option box_count=1
run headline --steps 200 --warmup 30
timeout 1200 python tests/tools/stress_sweep_vs_reference.py 10000 6000 > $O/stress_sweep_vs_reference.txt 2>&1; tail -2 $O/stress_sweep_vs_reference.txt | cut -c1-600
