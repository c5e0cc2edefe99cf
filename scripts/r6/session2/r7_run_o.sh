O=gpurun_out/r7p; mkdir -p $O
python -m pytest tests/test_parity_gpu.py tests/test_reference_golden.py -q -m gpu -x 2>&1 | tail -2 | tee $O/pytest_parity.txt
bash scripts/ab_run.sh $O "--no-camera-sequence --no-config-legs --steps 300 --warmup 50" bitonic
echo "== 500k"; bash scripts/ab_run.sh ${O}_x "--gaussians 500000 --no-camera-sequence --no-config-legs --steps 150 --warmup 30" bitonic
echo "== 2M"; bash scripts/ab_run.sh ${O}_y "--gaussians 2000000 --no-camera-sequence --no-config-legs --steps 150 --warmup 30" bitonic
