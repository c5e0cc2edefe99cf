mkdir -p gpurun_out/r2c3
timeout 1800 python -m pytest tests -m gpu -x -q -s 2>&1 | grep -v "^$" | tail -40 > gpurun_out/r2c3/pytest_gpu.log
tail -12 gpurun_out/r2c3/pytest_gpu.log
B="python bench.py --steps 200 --warmup 30 --no-cpu-baseline"
run() { n=$1; shift; timeout 300 $B "$@" > gpurun_out/r2c3/bench_$n.log 2>&1; tail -1 gpurun_out/r2c3/bench_$n.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n', d['value'], d['forward_fps'], d['stages_ms']['render_forward'], d['stages_ms']['render_backward'], d['stages_ms']['preprocess_backward'])"; }
run default
run default_norec --option grad_record=0
for n in base hoist flat hoistflat; do
  WG_RASTERIZER_LIB=$PWD/wild-gaussians_amd/build/fwd_$n/libwg_rasterizer.so run fwd_$n
done
