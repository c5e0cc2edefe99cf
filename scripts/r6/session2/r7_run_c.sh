# round 6, second session: K9's reduction: LDS (default) vs first butterfly stage + LDS (ldsred2) vs butterfly
O=gpurun_out/r7c; mkdir -p $O
WG_RASTERIZER_LIB=$PWD/wild-gaussians_amd/build/ldsred2/libwg_rasterizer.so python -m pytest tests/test_parity_gpu.py -q -m gpu -x -k "backward or grad or config2" 2>&1 | tail -3 | tee $O/pytest_ldsred2.txt
bash scripts/ab_run.sh $O "--no-camera-sequence --no-config-legs --steps 300 --warmup 50" ldsred2 butterfly
