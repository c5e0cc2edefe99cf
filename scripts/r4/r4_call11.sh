#!/bin/bash
O=gpurun_out/r4c11; mkdir -p $O
python -m pytest tests/test_torch_binding.py -q --timeout 900 > $O/pytest_binding.log 2>&1; echo "pytest rc $?" | tee $O/summary.txt; tail -5 $O/pytest_binding.log
python scripts/diag_host_wait.py > $O/host_wait_1M.json 2> $O/host_wait.err; python - $O/host_wait_1M.json <<'PY' | tee -a $O/summary.txt
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
for f in d["flows"]: print(f["flow"][:60].ljust(62), "host", f["host_call_ms_p50"], "gpu", f["gpu_ms_p50"])
print(d.get("compiled_binding",""))
PY
