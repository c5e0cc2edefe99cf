mkdir -p gpurun_out/r3n
(time python -m pytest tests/test_parity_gpu.py -m gpu -q -k "deterministic or geometry_reuse or gradient_record or fixed_capacity") > gpurun_out/r3n/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3n/pytest.log
tail -5 gpurun_out/r3n/pytest.log
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats -d gpurun_out/r3n/trace_det -o t -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-profile --option deterministic_backward=1 > gpurun_out/r3n/trace_det.log 2>&1
python scripts/rocpd_summary.py gpurun_out/r3n/trace_det/t_results.db > gpurun_out/r3n/kernel_trace_det.txt 2>&1
rm -rf gpurun_out/r3n/trace_det
cut -c1-150 gpurun_out/r3n/kernel_trace_det.txt | head -6
for rep in 1 2; do
python bench.py --steps 300 --warmup 50 --no-cpu-baseline > gpurun_out/r3n/headline.$rep.json 2>/dev/null
python bench.py --steps 300 --warmup 50 --no-cpu-baseline --option deterministic_backward=1 > gpurun_out/r3n/headline_det.$rep.json 2>/dev/null
done
python bench.py --steps 200 --warmup 30 --no-cpu-baseline --scale-mult 3 --option deterministic_backward=1 > gpurun_out/r3n/x3_det.json 2>/dev/null
python bench.py --steps 200 --warmup 30 --no-cpu-baseline --scale-mult 3 > gpurun_out/r3n/x3.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3n/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); s=d.get('stages_ms',{})
        print(f.split('/')[-1], 'value',d['value'],'ms',d['ms_per_step'], 'bwd', s.get('render_backward'))
    except Exception as e: print(f,'FAIL',e)
PY
