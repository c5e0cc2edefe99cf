#!/bin/bash
# round 5, second session (the container was re-created: every binary rebuilt from the committed sources): one GPU-box call that
# re-checks the rebuilt tree and re-collects the stamped profiles on it.  gpurun -- 'bash scripts/r5/r5_reentry.sh'
# Steps in order of priority, each under its own timeout; the optional tail (config 5) only runs while the call's time allows.
O=gpurun_out/r5b; mkdir -p $O
T0=$(date +%s); el() { echo $(( $(date +%s) - T0 )); }
log() { echo "[$(el) s] $*" | tee -a $O/steps.log; }
log "pytest -m gpu"
(timeout 420 python -m pytest tests -m gpu -q) > $O/pytest_gpu.log 2>&1; log "pytest rc=$? : $(tail -1 $O/pytest_gpu.log)"
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; log "smoke rc=$? : $(tail -1 $O/smoke.log)"
timeout 240 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; log "driver-command bench rc=$?"
timeout 420 bash scripts/profile_gpu.sh r5b_prof_headline > $O/profile_headline.log 2>&1; log "headline profile rc=$?"
[ -s gpurun_out/r5b_prof_headline/pmc_traffic.json ] && cp gpurun_out/r5b_prof_headline/pmc_traffic.json profiles/pmc_traffic.json
timeout 180 python tests/tools/count_pairs.py > $O/pair_counts.json 2> $O/pair_counts.err; log "pair counts rc=$?"
[ -s $O/pair_counts.json ] && cp $O/pair_counts.json profiles/pair_counts.json
timeout 300 python bench.py > $O/bench_final.json 2> $O/bench_final.err; log "default bench rc=$?"
WG_DIST_BACKEND=gloo timeout 240 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline --no-camera-sequence > $O/bench_gpus2_one_device_gloo.json 2> $O/bench_gpus2.err; log "two ranks on one device over gloo rc=$?"
if [ $(el) -lt 560 ]; then
  WORKLOAD="10000000 Gaussians, 3840x2160, sh" timeout 420 bash scripts/profile_gpu.sh r5b_prof_config5 --gaussians 10000000 --width 3840 --height 2160 --forward-only > $O/profile_config5.log 2>&1; log "config-5 profile rc=$?"
  [ -s gpurun_out/r5b_prof_config5/pmc_traffic.json ] && cp gpurun_out/r5b_prof_config5/pmc_traffic.json profiles/pmc_traffic_config5.json
fi
if [ $(el) -lt 780 ]; then
  timeout 200 python tests/tools/count_pairs.py --gaussians 10000000 --width 3840 --height 2160 --forward-only --oracle-gaussians 2000000 > $O/pair_counts_config5.json 2> $O/pair_counts_c5.err; log "config-5 pair counts rc=$?"
  [ -s $O/pair_counts_config5.json ] && cp $O/pair_counts_config5.json profiles/pair_counts_config5.json
fi
if [ $(el) -lt 860 ]; then
  timeout 200 python bench.py --gaussians 10000000 --width 3840 --height 2160 --forward-only --steps 100 --warmup 10 > $O/bench_config5_10M_4K_forward.json 2> $O/bench_config5.err; log "config-5 bench rc=$?"
fi
python - $O <<'PY' | tee $O/summary.txt
import json,sys,glob
for f in sorted(glob.glob(sys.argv[1]+"/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        if "value" in d:
            r=d.get("roofline") or {}
            print(f"{f.split('/')[-1]:40s} {d['value']:8.1f} {d['unit']} n_gpus {d['n_gpus']} fwd {d.get('forward_fps',0):8.1f} fps ms/step {d['ms_per_step']} stages {d.get('stages_ms')}")
            print("      roofline", {k:r.get(k) for k in ("kernel","frac","avg_launch_ms","traffic","traffic_over_algorithmic_bytes")}, "stamps", {k:d.get('library',{}).get(k) for k in ("kernel_source_sha","device_code_sha")}, d.get("stage_rooflines_note"))
            for k in ("parity","cpu_baseline","collective_backend","per_rank_ms_per_step","speedup_vs_reference_on_this_gpu"):
                if k in d: print("     ",k,json.dumps(d[k])[:500])
        else:
            print(f.split('/')[-1], json.dumps(d)[:400])
    except Exception as e: print(f, "FAILED", e)
PY
log "done"
