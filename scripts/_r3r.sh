mkdir -p gpurun_out/r3r
(time python -m pytest tests -m gpu -q) > gpurun_out/r3r/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3r/pytest.log
tail -6 gpurun_out/r3r/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
scripts/profile_gpu.sh r3_prof_headline > gpurun_out/r3_prof_headline.log 2>&1
WORKLOAD="10000000 Gaussians, 3840x2160, sh" scripts/profile_gpu.sh r3_prof_config5 --gaussians 10000000 --width 3840 --height 2160 --forward-only > gpurun_out/r3_prof_config5.log 2>&1
cp gpurun_out/r3_prof_headline/pmc_traffic.json profiles/pmc_traffic.json
cp gpurun_out/r3_prof_config5/pmc_traffic.json profiles/pmc_traffic_config5.json
python bench.py > gpurun_out/r3r/bench_final.json 2> gpurun_out/r3r/bench_final.err
python bench.py --gaussians 10000000 --width 3840 --height 2160 --forward-only --steps 100 --warmup 20 > gpurun_out/r3r/bench_config5.json 2> gpurun_out/r3r/bench_config5.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3r/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], 'value',d['value'],'fwd',d.get('forward_fps'),'ms',d['ms_per_step'], 'roofline', {k:d.get('roofline',{}).get(k) for k in ('bound','kernel','frac')}, d.get('speedup_vs_reference_on_this_gpu'), d['stages_ms'])
    except Exception as e: print(f,'FAIL',e)
PY
