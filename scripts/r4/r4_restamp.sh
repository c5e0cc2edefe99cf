#!/bin/bash
# restamp the PMC profiles to the final sources and re-take the headline lines
O=gpurun_out/r4final2; mkdir -p $O
bash scripts/profile_gpu.sh r4_prof_headline > $O/profile_headline.log 2>&1
WORKLOAD="10000000 Gaussians, 3840x2160, sh" bash scripts/profile_gpu.sh r4_prof_config5 --gaussians 10000000 --width 3840 --height 2160 --forward-only > $O/profile_config5.log 2>&1
cp gpurun_out/r4_prof_headline/pmc_traffic.json profiles/pmc_traffic.json; cp gpurun_out/r4_prof_config5/pmc_traffic.json profiles/pmc_traffic_config5.json
python bench.py > $O/bench_final.json 2> $O/bench_final.err
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
python bench.py --gaussians 10000000 --width 3840 --height 2160 --forward-only --steps 100 --warmup 10 > $O/bench_config5_10M_4K_forward.json 2> $O/bench_config5.err
python -m pytest tests -m gpu -q --timeout 900 > $O/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python - $O <<'PY'
import json,sys,glob
for f in sorted(glob.glob(sys.argv[1]+"/*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d.get("roofline",{})
    print(f.split('/')[-1], d["value"], "fwd", d.get("forward_fps"), "ms", d["ms_per_step"], d.get("step_ms_quantiles"), {k:r.get(k) for k in ("bound","frac","valu_busy")}, r.get("hbm",{}).get("by_survey_8d_bytes",{}).get("frac"), d["library"]["kernel_source_sha"])
PY
