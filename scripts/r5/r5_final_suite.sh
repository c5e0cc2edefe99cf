#!/bin/bash
# round 5, second session: the whole GPU suite + smoke on the final tree, under the default (compiled) binding and under the ctypes binding
O=gpurun_out/r5g; mkdir -p $O
T0=$(date +%s); el() { echo $(( $(date +%s) - T0 )); }
(timeout 420 python -m pytest tests -m gpu -q) > $O/pytest_gpu_compiled_binding.log 2>&1; echo "[$(el) s] pytest (compiled binding) rc=$? : $(tail -1 $O/pytest_gpu_compiled_binding.log)" | tee -a $O/steps.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "[$(el) s] smoke rc=$? : $(tail -1 $O/smoke.log)" | tee -a $O/steps.log
(WG_BINDING=ctypes timeout 420 python -m pytest tests -m gpu -q) > $O/pytest_gpu_ctypes_binding.log 2>&1; echo "[$(el) s] pytest (ctypes binding) rc=$? : $(tail -1 $O/pytest_gpu_ctypes_binding.log)" | tee -a $O/steps.log
