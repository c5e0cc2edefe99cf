#!/bin/bash
# round 4, GPU call 2: late-wave parking with per-band L2 counters -- correctness, then A/B at several thresholds
O=gpurun_out/r4c2; mkdir -p $O
timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -q -x -k "parking" > $O/pytest_parking.log 2>&1; echo "pytest parking rc $?" | tee -a $O/summary.txt; tail -5 $O/pytest_parking.log
B="--steps 300 --warmup 30 --no-cpu-baseline"
for rep in 1 2; do
  for pct in 0 70 80 88 94; do
    timeout 300 python bench.py $B --option forward_parking=$pct > $O/park$pct.$rep.json 2> $O/park$pct.$rep.err
  done
done
python - $O <<'PY' | tee -a $O/summary.txt
import json,sys,glob
for f in sorted(glob.glob(sys.argv[1]+"/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); s=d.get("stages_ms",{})
        print(f"{f.split('/')[-1]:22s} train {d['value']:8.1f} it/s fwd {d.get('forward_fps',0):8.1f} fps render_fwd {s.get('render_forward',0):.4f} render_bwd {s.get('render_backward',0):.4f}")
    except Exception as e: print(f, "FAILED", e)
PY
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --option forward_parking=88 > $O/trace.log 2>&1
python scripts/rocpd_summary.py $O/trace/t_results.db > $O/kernel_trace_summary_park88.txt 2>&1; rm -rf $O/trace
cut -c1-150 $O/kernel_trace_summary_park88.txt | head -16
