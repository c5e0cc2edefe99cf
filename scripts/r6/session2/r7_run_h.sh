# round 6, second session: non-temporal SH streams over the number of Gaussians (where does the gain at 1 M turn into the loss at 10 M?); the new controller test
python -m pytest tests/test_parity_gpu.py -q -m gpu -x -k "near_aim or near_far_split" 2>&1 | tail -3
for P in 500000 2000000 3000000 5000000; do
echo "== $P Gaussians, 1080p, SH 3, fwd+bwd"
bash scripts/ab_run.sh gpurun_out/r7h_$P "--gaussians $P --no-camera-sequence --no-config-legs --steps 150 --warmup 30" ntstream
done
