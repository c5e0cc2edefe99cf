#!/bin/bash
# round 5, second session: the whole GPU suite + smoke on the final tree (both bindings for the deterministic / threaded tests)
O=gpurun_out/r5e; mkdir -p $O
T0=$(date +%s); el() { echo $(( $(date +%s) - T0 )); }
(timeout 420 python -m pytest tests -m gpu -q) > $O/pytest_gpu.log 2>&1; echo "[$(el) s] pytest rc=$? : $(tail -1 $O/pytest_gpu.log)" | tee -a $O/steps.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "[$(el) s] smoke rc=$? : $(tail -1 $O/smoke.log)" | tee -a $O/steps.log
(WG_BINDING=ctypes timeout 300 python -m pytest tests/test_parity_gpu.py tests/test_native_driver.py -m gpu -q -k "deterministic or concurrent or stream or driver") > $O/pytest_ctypes_subset.log 2>&1
echo "[$(el) s] ctypes-binding subset rc=$? : $(tail -1 $O/pytest_ctypes_subset.log)" | tee -a $O/steps.log
timeout 240 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; echo "[$(el) s] driver-command bench rc=$?" | tee -a $O/steps.log
python - <<'PY' | tee -a $O/steps.log
import json
d=json.loads(open('gpurun_out/r5e/bench_driver_cmd.json').read().strip().splitlines()[-1]); r=d['roofline']
print(d['value'], d['unit'], d['ms_per_step'], 'ms; roofline', {k:r.get(k) for k in ('kernel','frac','traffic','traffic_over_algorithmic_bytes')}, {k:d['library'].get(k) for k in ('kernel_source_sha','device_code_sha')}, d.get('stage_rooflines_note'))
PY
