#!/bin/bash
O=gpurun_out/r4c6; mkdir -p $O
python -m pytest tests -m gpu -q -x --timeout 900 > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" | tee $O/summary.txt; tail -6 $O/pytest_gpu.log
python scripts/diag_idle_ramp.py > $O/diag_idle_ramp.json 2> $O/diag_idle_ramp.err; cat $O/diag_idle_ramp.json | tee -a $O/summary.txt
for rep in 1 2; do
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/driver_cmd.$rep.json 2> $O/driver_cmd.$rep.err
python - $O/driver_cmd.$rep.json <<'PY' | tee -a $O/summary.txt
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("driver cmd: value", d["value"], "ms_per_step", d["ms_per_step"], "quantiles", d.get("step_ms_quantiles"), "host", d.get("timed_region_host_ms"))
PY
done
