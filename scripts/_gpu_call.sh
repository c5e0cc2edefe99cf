cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2c9; mkdir -p $O
timeout 900 python bench.py > $O/bench_final.log 2>&1; tail -1 $O/bench_final.log | cut -c1-300
B="python bench.py --no-cpu-baseline"
run() { n=$1; shift; timeout 600 $B "$@" > $O/bench_$n.log 2>&1; tail -1 $O/bench_$n.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n', d['value'], d['forward_fps'], d['stages_ms'], d['roofline']['frac'], d['roofline']['frac_basis'])" || tail -5 $O/bench_$n.log; }
run config5 --steps 50 --warmup 5 --gaussians 10000000 --width 3840 --height 2160 --forward-only
run config2_500k --steps 300 --warmup 30 --gaussians 500000
run 3M_1600x1200 --steps 200 --warmup 20 --gaussians 3000000 --width 1600 --height 1200 --colors precomp
run dense_x3 --steps 200 --warmup 20 --scale-mult 3
run gpus2 --gpus 2 --steps 100 --warmup 10
