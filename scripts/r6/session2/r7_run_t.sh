# round 6, second session: the lazy sort's front extraction appends in one LDS round trip per trip (default) vs one per slot (extract_before)
O=gpurun_out/r7t; mkdir -p $O
python -m pytest tests/test_parity_gpu.py -q -m gpu -x -k "lazy or near or dense or split or fixup" 2>&1 | tail -2
echo "== 10M 4K"; bash scripts/ab_run.sh ${O}_c5 "--gaussians 10000000 --width 3840 --height 2160 --forward-only --no-camera-sequence --steps 200 --warmup 100" extract_before
echo "== dense x3"; bash scripts/ab_run.sh ${O}_x3 "--scale-mult 3 --no-camera-sequence --no-config-legs --steps 200 --warmup 30" extract_before
echo "== 3M 1600x1200 precomp"; bash scripts/ab_run.sh ${O}_x "--gaussians 3000000 --width 1600 --height 1200 --colors precomp --no-camera-sequence --no-config-legs --steps 150 --warmup 30" extract_before
