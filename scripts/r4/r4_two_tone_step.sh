#!/bin/bash
# the restated WildGaussians step at 3 M Gaussians, every opt-in: two toned calls vs ONE two-tone call, activations stand-alone vs in-kernel
mkdir -p gpurun_out/r4_two_tone
COMMON="--gaussians 3000000 --width 1600 --height 1200 --steps 30 --warmup 8 --in-kernel-tone --fused-loss --wg-adam --densification-stats fused --tall-linear"
for rep in 1 2; do
timeout 300 python scripts/bench_wildgaussians_step.py $COMMON --fused-activations > gpurun_out/r4_two_tone/two_calls_fused_act_$rep.json 2> gpurun_out/r4_two_tone/err_a_$rep.log
timeout 300 python scripts/bench_wildgaussians_step.py $COMMON --fused-activations --two-tone-call > gpurun_out/r4_two_tone/one_call_fused_act_$rep.json 2> gpurun_out/r4_two_tone/err_b_$rep.log
timeout 300 python scripts/bench_wildgaussians_step.py $COMMON --in-kernel-activations --two-tone-call > gpurun_out/r4_two_tone/one_call_in_kernel_act_$rep.json 2> gpurun_out/r4_two_tone/err_c_$rep.log
done
tail -n 2 gpurun_out/r4_two_tone/*.json gpurun_out/r4_two_tone/err_*_1.log
