#!/bin/bash
# kernel trace of the reference's REAL train_iteration at 3 M Gaussians with apply_optins(render_edit="two_tone")
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4_real_trace; mkdir -p $O
timeout 500 rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python scripts/bench_wildgaussians_step.py --real-caller --gaussians 3000000 --width 1600 --height 1200 --steps 10 --warmup 4 --optins --two-tone-edit > $O/run.log 2> $O/run.err
python scripts/rocpd_summary.py $O/trace/t_results.db > $O/kernel_trace_summary.txt 2>&1
rm -rf $O/trace
head -45 $O/kernel_trace_summary.txt | cut -c1-170
tail -1 $O/run.log | cut -c1-300
