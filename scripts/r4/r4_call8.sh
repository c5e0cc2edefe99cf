#!/bin/bash
# round 4, GPU call 8: profiles on the round's kernels (kernel trace + PMC passes, headline and config 5), the FMA-rate probe, raw-mode test
O=gpurun_out/r4c8; mkdir -p $O
python -m pytest tests/test_activations.py tests/test_real_caller.py -m gpu -q -x --timeout 900 > $O/pytest_new.log 2>&1; echo "pytest rc $?" | tee $O/summary.txt; tail -4 $O/pytest_new.log
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 scripts/pk_probe.hip -o $O/pk_probe && $O/pk_probe > $O/pk_probe.txt 2>&1; cat $O/pk_probe.txt | tee -a $O/summary.txt; rm -f $O/pk_probe
bash scripts/profile_gpu.sh r4_prof_headline > $O/profile_headline.log 2>&1; tail -20 $O/profile_headline.log
WORKLOAD="10000000 Gaussians, 3840x2160, sh" bash scripts/profile_gpu.sh r4_prof_config5 --gaussians 10000000 --width 3840 --height 2160 --forward-only > $O/profile_config5.log 2>&1; tail -12 $O/profile_config5.log
