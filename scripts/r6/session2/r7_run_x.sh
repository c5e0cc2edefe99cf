# round 6, second session: lazy colour of split frames (default) vs every visible Gaussian coloured by the per-Gaussian kernel (lazy_before = commit 6536ccc)
O=gpurun_out/r7x; mkdir -p $O
python -m pytest tests/test_parity_gpu.py tests/test_reference_golden.py -q -m gpu -x 2>&1 | tail -2
echo "== 10M 4K"; bash scripts/ab_run.sh ${O}_c5 "--gaussians 10000000 --width 3840 --height 2160 --forward-only --no-camera-sequence --steps 200 --warmup 100" lazy_before
echo "== 3M 1080p SH fwd+bwd"; bash scripts/ab_run.sh ${O}_x "--gaussians 3000000 --no-camera-sequence --no-config-legs --steps 150 --warmup 30" lazy_before
echo "== 1M dense x3"; bash scripts/ab_run.sh ${O}_x3 "--scale-mult 3 --no-camera-sequence --no-config-legs --steps 200 --warmup 30" lazy_before
