#!/bin/bash
# Everything a round ends with, in one GPU-box call:  the GPU suite, smoke(), the rocprofv3 kernel trace + PMC passes of the headline
# and config-5 workloads (which stamp profiles/pmc_traffic*.json to the kernel sources), then the bench lines that read those stamps.
# usage (from the repo root, through gpurun):  scripts/round_checks.sh <tag>      -> gpurun_out/<tag>/, gpurun_out/<tag>_prof_*/
set -u
TAG=${1:-round}
mkdir -p gpurun_out/$TAG
(time python -m pytest tests -m gpu -q) > gpurun_out/$TAG/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/$TAG/pytest.log
tail -6 gpurun_out/$TAG/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
scripts/profile_gpu.sh ${TAG}_prof_headline > gpurun_out/${TAG}_prof_headline.log 2>&1
WORKLOAD="10000000 Gaussians, 3840x2160, sh" scripts/profile_gpu.sh ${TAG}_prof_config5 --gaussians 10000000 --width 3840 --height 2160 --forward-only > gpurun_out/${TAG}_prof_config5.log 2>&1
cp gpurun_out/${TAG}_prof_headline/pmc_traffic.json profiles/pmc_traffic.json
cp gpurun_out/${TAG}_prof_config5/pmc_traffic.json profiles/pmc_traffic_config5.json
python bench.py > gpurun_out/$TAG/bench_final.json 2> gpurun_out/$TAG/bench_final.err
python bench.py --gaussians 10000000 --width 3840 --height 2160 --forward-only --steps 100 --warmup 20 > gpurun_out/$TAG/bench_config5.json 2> gpurun_out/$TAG/bench_config5.err
python - $TAG <<'PY'
import json, glob, sys
for f in sorted(glob.glob(f'gpurun_out/{sys.argv[1]}/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], 'value', d['value'], 'fwd', d.get('forward_fps'), 'ms', d['ms_per_step'],
              {k: d.get('roofline', {}).get(k) for k in ('bound', 'kernel', 'frac')}, d.get('speedup_vs_reference_on_this_gpu'))
    except Exception as e:  # noqa: BLE001
        print(f, 'FAIL', e)
PY
