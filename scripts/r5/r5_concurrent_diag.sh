#!/bin/bash
# round 5: where do the concurrent callers of tests/native/c_abi_driver.cpp deviate?  (one deviating deterministic call in the GPU suite of
# the re-entry run.)  Runs the driver's concurrent block under its diagnostic knobs, product library and the cached-scratch variant.
O=gpurun_out/r5c; mkdir -p $O
T0=$(date +%s); el() { echo $(( $(date +%s) - T0 )); }
D=wild-gaussians_amd/build/c_abi_driver; DC=wild-gaussians_amd/build/cached/c_abi_driver
run() {  # name, runs, driver, env...
  local name=$1 runs=$2 drv=$3; shift 3
  local bad=0
  for r in $(seq 1 $runs); do
    [ $(el) -gt ${LIMIT:-330} ] && { echo "$name: out of time after $((r-1)) runs" | tee -a $O/conc.log; break; }
    env WG_DRV_VERBOSE=1 "$@" timeout 90 $drv 200000 1280 720 > $O/run.out 2> $O/run.err; rc=$?
    [ $rc -ne 0 ] && bad=$((bad+1))
    echo "== $name run $r rc=$rc [$(el) s]" >> $O/conc.log; grep -v "^ok" $O/run.err | head -40 >> $O/conc.log
  done
  echo "## $name: $bad of $runs runs deviated [$(el) s]" | tee -a $O/conc.log
}
run product_default 6 $D WG_DRV_REPEAT=4
run product_nobody_deterministic 4 $D WG_DRV_REPEAT=4 WG_DRV_DET_MASK=0
run product_all_deterministic 4 $D WG_DRV_REPEAT=4 WG_DRV_DET_MASK=7
run product_one_thread_deterministic 3 $D WG_DRV_REPEAT=8 WG_DRV_THREADS=1 WG_DRV_DET_MASK=1
run product_keep_buffers 4 $D WG_DRV_REPEAT=4 WG_DRV_KEEP=1
run cached_default 6 $DC WG_DRV_REPEAT=4
run cached_all_deterministic 4 $DC WG_DRV_REPEAT=4 WG_DRV_DET_MASK=7
grep "^##" $O/conc.log
