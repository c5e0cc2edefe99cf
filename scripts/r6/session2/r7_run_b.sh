# round 6, second session: K9's reduction through LDS (default) vs the butterfly (variant build), same box
O=gpurun_out/r7b; mkdir -p $O
python -m pytest tests/test_parity_gpu.py tests/test_reference_golden.py -q -m gpu -x 2>&1 | tail -5 | tee $O/pytest_parity.txt
bash scripts/ab_run.sh $O "--no-camera-sequence --no-config-legs --steps 300 --warmup 50" butterfly
