# round 6, second session: scatter's list appends in one LDS round trip per pass (default; tile_count: one per slot) vs the library before both (count_before)
O=gpurun_out/r7u; mkdir -p $O
python -m pytest tests/test_parity_gpu.py -q -m gpu -x 2>&1 | tail -2
bash scripts/ab_run.sh $O "--no-camera-sequence --no-config-legs --steps 300 --warmup 50" count_before
echo "== 10M 4K"; bash scripts/ab_run.sh ${O}_c5 "--gaussians 10000000 --width 3840 --height 2160 --forward-only --no-camera-sequence --steps 200 --warmup 100" count_before
echo "== 3M 1600x1200 precomp"; bash scripts/ab_run.sh ${O}_x "--gaussians 3000000 --width 1600 --height 1200 --colors precomp --no-camera-sequence --no-config-legs --steps 150 --warmup 30" count_before
