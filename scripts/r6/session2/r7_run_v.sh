# round 6, second session: band lists kept in a near and a far part (default) vs one list (lists_before = commit 28d461f)
O=gpurun_out/r7w; mkdir -p $O
python -m pytest tests/test_parity_gpu.py -q -m gpu -x 2>&1 | tail -2
echo "== 10M 4K"; bash scripts/ab_run.sh ${O}_c5 "--gaussians 10000000 --width 3840 --height 2160 --forward-only --no-camera-sequence --steps 200 --warmup 100" lists_before
echo "== 3M 1600x1200 precomp"; bash scripts/ab_run.sh ${O}_x "--gaussians 3000000 --width 1600 --height 1200 --colors precomp --no-camera-sequence --no-config-legs --steps 150 --warmup 30" lists_before
echo "== 3M 1600x1200 precomp x2"; bash scripts/ab_run.sh ${O}_x2 "--gaussians 3000000 --width 1600 --height 1200 --colors precomp --scale-mult 2 --no-camera-sequence --no-config-legs --steps 150 --warmup 30" lists_before
