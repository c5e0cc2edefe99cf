#!/bin/bash
# after scripts/r6/r6_final.sh (its outputs merged back into gpurun_out/): copy what is to be judged into profiles/ (tracked)
set -e
mkdir -p profiles/r6_bench profiles/r6_prof_headline profiles/r6_prof_config5
cp gpurun_out/r6final/*.json gpurun_out/r6final/summary.txt gpurun_out/r6final/stream_overlap_views8_streams2.txt gpurun_out/r6final/malloc_async_lost_stores.txt profiles/r6_bench/ 2>/dev/null || true
cp gpurun_out/r6final/near_trace_synchronized.txt gpurun_out/r6final/near_trace_pipelined.txt profiles/r6_bench/ 2>/dev/null || true
cp gpurun_out/r6final/pytest_gpu.log profiles/r6_bench/pytest_gpu_final.log
cp gpurun_out/r6final/smoke.log profiles/r6_bench/smoke_final.log
for d in headline config5; do
  for f in kernel_trace_summary.txt pmc_summary.txt pmc_traffic.json bench_under_trace.json; do cp gpurun_out/r6_prof_$d/$f profiles/r6_prof_$d/ 2>/dev/null || true; done
done
cp gpurun_out/r6final_cal/pmc_calibration.json profiles/r6/pmc_calibration.json 2>/dev/null || true
ls profiles/r6_bench | wc -l
# (r6_final.sh copies these on the GPU box for its own later legs; only gpurun_out/ travels back)
cp gpurun_out/r6_prof_headline/pmc_traffic.json profiles/pmc_traffic.json
cp gpurun_out/r6_prof_config5/pmc_traffic.json profiles/pmc_traffic_config5.json
cp gpurun_out/r6final/pair_counts.json profiles/pair_counts.json
cp gpurun_out/r6final/pair_counts_config5.json profiles/pair_counts_config5.json
