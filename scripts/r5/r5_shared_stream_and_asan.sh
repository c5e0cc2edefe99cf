#!/bin/bash
# round 5, second session: callers that feed ONE stream from several host threads (the leased scratch blocks' last case), and the driver under
# AddressSanitizer (host side of the library) with the concurrent block in its variants.
O=gpurun_out/r5f; mkdir -p $O
T0=$(date +%s); el() { echo $(( $(date +%s) - T0 )); }
run() {  # name, runs, driver, env...
  local name=$1 runs=$2 drv=$3; shift 3
  local bad=0
  for r in $(seq 1 $runs); do
    env WG_DRV_VERBOSE=1 "$@" timeout 120 $drv 200000 1280 720 > $O/run.out 2> $O/run.err; rc=$?
    [ $rc -ne 0 ] && bad=$((bad+1))
    echo "== $name run $r rc=$rc [$(el) s]" >> $O/conc.log; grep -v "^ok" $O/run.err | head -30 >> $O/conc.log
  done
  echo "## $name: $bad of $runs runs deviated [$(el) s]" | tee -a $O/conc.log
}
D=wild-gaussians_amd/build/c_abi_driver; DA=wild-gaussians_amd/build/asan/c_abi_driver
run one_stream_three_threads_caller1_deterministic 6 $D WG_DRV_REPEAT=4 WG_DRV_SHARED_STREAM=1
run one_stream_three_threads_all_deterministic 6 $D WG_DRV_REPEAT=4 WG_DRV_SHARED_STREAM=1 WG_DRV_DET_MASK=7
run one_stream_nobody_deterministic 3 $D WG_DRV_REPEAT=4 WG_DRV_SHARED_STREAM=1 WG_DRV_DET_MASK=0
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:abort_on_error=0:protect_shadow_gap=0
run asan_default 2 $DA WG_DRV_REPEAT=2
run asan_all_deterministic 2 $DA WG_DRV_REPEAT=2 WG_DRV_DET_MASK=7
run asan_one_stream_all_deterministic 2 $DA WG_DRV_REPEAT=2 WG_DRV_DET_MASK=7 WG_DRV_SHARED_STREAM=1
timeout 120 $DA > $O/asan_small.out 2> $O/asan_small.err; echo "asan small scene rc=$? : $(tail -1 $O/asan_small.out)" | tee -a $O/conc.log
echo "asan errors: $(cat $O/conc.log $O/asan_small.err | grep -c 'ERROR: AddressSanitizer')" | tee -a $O/conc.log
echo "[$(el) s] done" | tee -a $O/conc.log
