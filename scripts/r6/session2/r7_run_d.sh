# round 6, second session: K9 reduction variants 2 / 3; near-aim trace in bench.py's order of passes
O=gpurun_out/r7d; mkdir -p $O
WG_RASTERIZER_LIB=$PWD/wild-gaussians_amd/build/ldsred3/libwg_rasterizer.so python -m pytest tests/test_parity_gpu.py -q -m gpu -x -k "backward or grad or config2" 2>&1 | tail -3 | tee $O/pytest_ldsred3.txt
bash scripts/ab_run.sh $O "--no-camera-sequence --no-config-legs --steps 300 --warmup 50" ldsred2 ldsred3
NEAR_TRACE_BENCH_LIKE=1 python scripts/r6/near_trace.py 1500 > $O/near_trace_bench_like.txt 2> $O/near_trace.err; tail -3 $O/near_trace.err; head -70 $O/near_trace_bench_like.txt
