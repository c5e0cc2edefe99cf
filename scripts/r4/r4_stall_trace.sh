#!/bin/bash
# kernel trace of a 12 000-step loop: are the slow steps gaps (GPU starved) or stretched kernels (GPU paused with work in hand)?
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4_diag; mkdir -p $O
timeout 400 rocprofv3 --kernel-trace -d $O/trace -o t -- python scripts/diag_step_blips.py 12000 > $O/blips_under_trace.json 2> $O/trace.err
ls $O/trace | head
python scripts/diag_stall_trace.py $O/trace/t_results.db > $O/stall_trace.json; cat $O/stall_trace.json | cut -c1-3000
tail -c 1500 $O/blips_under_trace.json
rm -rf $O/trace
