# round 6, second session: non-temporal SH streams (variant build) at the headline and at config 5
bash scripts/ab_run.sh gpurun_out/r7g "--no-camera-sequence --no-config-legs --steps 300 --warmup 50" ntstream
bash scripts/ab_run.sh gpurun_out/r7g_c5 "--gaussians 10000000 --width 3840 --height 2160 --forward-only --no-camera-sequence --steps 200 --warmup 60" ntstream
