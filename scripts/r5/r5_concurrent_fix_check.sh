#!/bin/bash
# round 5: the deterministic backward's scratch as leased hipMalloc blocks (api.hip: det_scratch_alloc) -- the concurrent callers of the
# torch-free driver again, many times, then the deterministic / threaded tests of the Python suite and what the mode costs.
O=gpurun_out/r5d; mkdir -p $O
T0=$(date +%s); el() { echo $(( $(date +%s) - T0 )); }
D=wild-gaussians_amd/build/c_abi_driver
run() {  # name, runs, env...
  local name=$1 runs=$2; shift 2
  local bad=0
  for r in $(seq 1 $runs); do
    env WG_DRV_VERBOSE=1 "$@" timeout 90 $D 200000 1280 720 > $O/run.out 2> $O/run.err; rc=$?
    [ $rc -ne 0 ] && bad=$((bad+1))
    echo "== $name run $r rc=$rc [$(el) s]" >> $O/conc.log; grep -v "^ok" $O/run.err | head -20 >> $O/conc.log
  done
  echo "## $name: $bad of $runs runs deviated [$(el) s]" | tee -a $O/conc.log
}
run leased_default 12 WG_DRV_REPEAT=4
run leased_all_deterministic 10 WG_DRV_REPEAT=4 WG_DRV_DET_MASK=7
run leased_five_threads_all_deterministic 5 WG_DRV_REPEAT=3 WG_DRV_THREADS=5 WG_DRV_DET_MASK=31
run leased_one_thread_deterministic 3 WG_DRV_REPEAT=8 WG_DRV_THREADS=1 WG_DRV_DET_MASK=1
(timeout 300 python -m pytest tests/test_native_driver.py tests/test_parity_gpu.py -m gpu -q -k "driver or deterministic or concurrent or stream") > $O/pytest_subset.log 2>&1
echo "[$(el) s] pytest subset rc=$? : $(tail -1 $O/pytest_subset.log)" | tee -a $O/conc.log
for i in 1 2; do
  timeout 200 python bench.py --no-cpu-baseline --no-camera-sequence --steps 300 --warmup 30 > $O/bench_atomic_$i.json 2> $O/b1.err
  timeout 200 python bench.py --no-cpu-baseline --no-camera-sequence --steps 300 --warmup 30 --option deterministic_backward=1 > $O/bench_deterministic_$i.json 2> $O/b2.err
done
python - $O <<'PY' | tee -a $O/conc.log
import json,sys,glob
for f in sorted(glob.glob(sys.argv[1]+"/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['value'], 'iter/s', d['ms_per_step'], 'ms', d.get('stages_ms'))
    except Exception as e: print(f, 'FAILED', e)
PY
echo "[$(el) s] done" | tee -a $O/conc.log
