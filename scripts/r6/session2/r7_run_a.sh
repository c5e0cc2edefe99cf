# round 6, second session: the K9 reduction A/B + a per-frame trace of the near / far split's adaptive aim at config 5
mkdir -p gpurun_out/r7a
bash scripts/r6/session2/r7_run_b.sh
python scripts/r6/near_trace.py 2500 > gpurun_out/r7a/near_trace_2500.txt 2> gpurun_out/r7a/near_trace.err; tail -3 gpurun_out/r7a/near_trace.err; head -60 gpurun_out/r7a/near_trace_2500.txt
