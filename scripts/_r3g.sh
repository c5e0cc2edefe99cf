mkdir -p gpurun_out/r3g
(time python -m pytest tests/test_parity_gpu.py -m gpu -q -k "deterministic or geometry_reuse or gradient_record") > gpurun_out/r3g/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3g/pytest.log
tail -6 gpurun_out/r3g/pytest.log
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats -d gpurun_out/r3g/trace_det -o t -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-profile --option deterministic_backward=1 > gpurun_out/r3g/trace_det.log 2>&1
python scripts/rocpd_summary.py gpurun_out/r3g/trace_det/t_results.db > gpurun_out/r3g/kernel_trace_det.txt 2>&1
rm -rf gpurun_out/r3g/trace_det
cut -c1-150 gpurun_out/r3g/kernel_trace_det.txt | head -8
for rep in 1 2; do
python bench.py --steps 300 --warmup 50 --no-cpu-baseline > gpurun_out/r3g/headline.$rep.json 2>/dev/null
python bench.py --steps 300 --warmup 50 --no-cpu-baseline --option deterministic_backward=1 > gpurun_out/r3g/headline_det.$rep.json 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3g/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); s=d.get('stages_ms',{})
        print(f.split('/')[-1], 'value',d['value'],'fwd',d.get('forward_fps'),'ms',d['ms_per_step'], 'bwd', s.get('render_backward'))
    except Exception as e: print(f,'FAIL',e)
PY
