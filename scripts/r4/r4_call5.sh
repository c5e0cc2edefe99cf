#!/bin/bash
# round 4, GPU call 5: full GPU suite (two-colour walk, hardened reuse, bench order) + the real caller's rasterizer share with one two-colour call
O=gpurun_out/r4c5; mkdir -p $O
python -m pytest tests -m gpu -q -x --timeout 900 > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" | tee $O/summary.txt; tail -12 $O/pytest_gpu.log
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/driver_cmd.json 2> $O/driver_cmd.err
python - $O/driver_cmd.json <<'PY' | tee -a $O/summary.txt
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("driver cmd: value", d["value"], "ms_per_step", d["ms_per_step"], "quantiles", d.get("step_ms_quantiles"), "host", d.get("timed_region_host_ms"))
PY
for mode in "" "--dual"; do
  for opt in 0 1; do
    n=real_caller_3M_reuse${opt}${mode// /}
    timeout 600 python scripts/bench_wildgaussians_step.py --real-caller --steps 10 --warmup 3 $mode --option geometry_reuse=$opt > $O/$n.json 2> $O/$n.err
    tail -1 $O/$n.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$n', {k:v for k,v in d.items() if 'ms' in k or 'share' in k})" | tee -a $O/summary.txt
  done
done
