mkdir -p gpurun_out/r3f
(time python -m pytest tests/test_parity_gpu.py tests/test_real_caller.py tests/test_reference_golden.py -m gpu -q --durations=3) > gpurun_out/r3f/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3f/pytest.log
tail -8 gpurun_out/r3f/pytest.log
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats -d gpurun_out/r3f/trace_det -o t -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-profile --option deterministic_backward=1 > gpurun_out/r3f/trace_det.log 2>&1
python scripts/rocpd_summary.py gpurun_out/r3f/trace_det/t_results.db > gpurun_out/r3f/kernel_trace_det.txt 2>&1
rm -rf gpurun_out/r3f/trace_det
cut -c1-150 gpurun_out/r3f/kernel_trace_det.txt | head -14
scripts/ab_run.sh gpurun_out/r3f/ab_sort "--steps 300 --warmup 50" sortold
