#!/bin/bash
# the reference's REAL train_iteration at 3 M Gaussians: run-time opt-ins, + the two-colour edit, + the two-tone edit
mkdir -p gpurun_out/r4_two_tone
timeout 300 python -m pytest tests/test_real_caller.py -q -x -k "two_tone or two_colour" 2>&1 | tail -15
COMMON="--real-caller --gaussians 3000000 --width 1600 --height 1200 --steps 20 --warmup 6 --optins"
timeout 300 python scripts/bench_wildgaussians_step.py $COMMON > gpurun_out/r4_two_tone/real_optins.json 2> gpurun_out/r4_two_tone/err_r1.log
timeout 300 python scripts/bench_wildgaussians_step.py $COMMON --two-colour-edit > gpurun_out/r4_two_tone/real_optins_two_colour_edit.json 2> gpurun_out/r4_two_tone/err_r2.log
timeout 300 python scripts/bench_wildgaussians_step.py $COMMON --two-tone-edit > gpurun_out/r4_two_tone/real_optins_two_tone_edit.json 2> gpurun_out/r4_two_tone/err_r3.log
tail -n 3 gpurun_out/r4_two_tone/real_*.json gpurun_out/r4_two_tone/err_r*.log
