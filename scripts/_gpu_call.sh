cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2c7; mkdir -p $O
B="python bench.py --no-cpu-baseline"
run() { n=$1; shift; timeout 600 $B "$@" > $O/bench_$n.log 2>&1; tail -1 $O/bench_$n.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n', d['value'], d['forward_fps'], d['stages_ms'])" || tail -5 $O/bench_$n.log; }
# MFMA reduction variant: parity, then time
WG_RASTERIZER_LIB=$PWD/wild-gaussians_amd/build/bwd_mfma/libwg_rasterizer.so timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "backward_gradient_parity or gradient_record" 2>&1 | tail -3
WG_RASTERIZER_LIB=$PWD/wild-gaussians_amd/build/bwd_mfma/libwg_rasterizer.so run bwd_mfma --steps 200 --warmup 30
run bwd_butterfly --steps 200 --warmup 30
run x4_auto_hint --steps 100 --warmup 10 --scale-mult 4
run x2_auto_hint --steps 100 --warmup 10 --scale-mult 2
run c5_npt1000 --steps 30 --warmup 5 --gaussians 10000000 --width 3840 --height 2160 --forward-only --option near_per_tile=1000
run c5_npt1400 --steps 30 --warmup 5 --gaussians 10000000 --width 3840 --height 2160 --forward-only --option near_per_tile=1400
