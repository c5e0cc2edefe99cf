#!/bin/bash
# Run on the GPU box (via gpurun): kernel trace + PMC passes for bench.py, summaries into gpurun_out/<tag>/.
# usage: [WORKLOAD="<P> Gaussians, <W>x<H>, <sh|precomp>[, scales x<m>]"] scripts/profile_gpu.sh <tag> [extra bench args]
# WORKLOAD is the key bench.py looks profiles/pmc_traffic.json up by (default: the headline scene).
set -u
TAG=${1:-prof}; shift || true
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
mkdir -p $OUT
BENCH="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-camera-sequence --no-config-legs $*"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- $BENCH > $OUT/trace.log 2>&1
python scripts/rocpd_summary.py $OUT/trace/t_results.db > $OUT/kernel_trace_summary.txt 2>&1
grep '"metric"' $OUT/trace.log > $OUT/bench_under_trace.json
PB="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-camera-sequence --no-config-legs $*"
i=0
for CS in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" \
          "SQ_INSTS_VMEM SQ_INSTS_VALU_TRANS_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
          "FETCH_SIZE GRBM_GUI_ACTIVE" \
          "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" \
          "SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INST_CYCLES_VALU SQ_INSTS_VALU_INT32"; do
  i=$((i+1))
  rocprofv3 --pmc $CS -d $OUT/pmc$i -o p -- $PB > $OUT/pmc$i.log 2>&1
  python scripts/rocpd_summary.py $OUT/pmc$i/p_results.db 2>&1 | grep -A100000 "PMC counters" | grep -E "wg::|PMC" >> $OUT/pmc_summary.txt
done
rm -rf $OUT/trace $OUT/pmc1 $OUT/pmc2 $OUT/pmc3 $OUT/pmc4 $OUT/pmc5
cat $OUT/kernel_trace_summary.txt | cut -c1-150 | head -16
python scripts/pmc_traffic.py $OUT/pmc_summary.txt "${WORKLOAD:-1000000 Gaussians, 1920x1080, sh}" > $OUT/pmc_traffic.json; head -c 600 $OUT/pmc_traffic.json
