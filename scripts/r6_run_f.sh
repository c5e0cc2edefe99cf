O=gpurun_out/r6f; mkdir -p $O
(time python -m pytest tests -m gpu -x -q) > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
python tests/tools/nonfinite_inputs.py 2>&1 | grep -v "libdrm" > $O/nonfinite.log; cat $O/nonfinite.log | cut -c1-200
bash scripts/r6/pmc_calibrate.sh r6f_cal > $O/cal.log 2>&1; tail -60 $O/cal.log
