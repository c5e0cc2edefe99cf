# round 6, second session: sh_colour_kernel with a list + coalesced block fetch through LDS (default) vs per-lane strided loads (colour_before = commit b9414f9)
O=gpurun_out/r7ad; mkdir -p $O
python -m pytest tests/test_parity_gpu.py -q -m gpu -x -k "lazy_colour or near_far or near_aim" 2>&1 | tail -2
echo "== 10M 4K"; bash scripts/ab_run.sh ${O}_c5 "--gaussians 10000000 --width 3840 --height 2160 --forward-only --no-camera-sequence --steps 200 --warmup 100" colour_before
echo "== 5M 1080p fwd+bwd"; bash scripts/ab_run.sh ${O}_5M "--gaussians 5000000 --no-camera-sequence --no-config-legs --steps 100 --warmup 30" colour_before
