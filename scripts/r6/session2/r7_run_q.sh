# round 6, second session: tile_count's per-band list appends in one LDS round trip (default) vs eight (count_before = the library before the change)
O=gpurun_out/r7r; mkdir -p $O
python -m pytest tests/test_parity_gpu.py -q -m gpu -x -k "band_list or near_far or lists or staged or scale" 2>&1 | tail -2
bash scripts/ab_run.sh $O "--gaussians 10000000 --width 3840 --height 2160 --forward-only --no-camera-sequence --steps 200 --warmup 100" count_before
echo "== 3M 1600x1200 precomp"; bash scripts/ab_run.sh ${O}_x "--gaussians 3000000 --width 1600 --height 1200 --colors precomp --no-camera-sequence --no-config-legs --steps 150 --warmup 30" count_before
