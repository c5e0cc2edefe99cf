# round 6, second session: the per-tile sort by depth bins (default) vs the bitonic network (variant build)
O=gpurun_out/r7n; mkdir -p $O
python -m pytest tests/test_parity_gpu.py tests/test_reference_golden.py -q -m gpu -x 2>&1 | tail -4 | tee $O/pytest_parity.txt
bash scripts/ab_run.sh $O "--no-camera-sequence --no-config-legs --steps 300 --warmup 50" bitonic
for a in "--gaussians 500000" "--scale-mult 3" "--gaussians 3000000 --width 1600 --height 1200 --colors precomp"; do
echo "== $a"
bash scripts/ab_run.sh ${O}_x "$a --no-camera-sequence --no-config-legs --steps 150 --warmup 30" bitonic
done
