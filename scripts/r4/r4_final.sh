#!/bin/bash
# round 4, final measurements on the round's final sources: suite, restamped profiles, bench lines of every BASELINE config, sweeps
O=gpurun_out/r4final; mkdir -p $O
python -m pytest tests -m gpu -q --timeout 900 > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" | tee $O/summary.txt; tail -3 $O/pytest_gpu.log | tee -a $O/summary.txt
bash scripts/profile_gpu.sh r4_prof_headline > $O/profile_headline.log 2>&1
WORKLOAD="10000000 Gaussians, 3840x2160, sh" bash scripts/profile_gpu.sh r4_prof_config5 --gaussians 10000000 --width 3840 --height 2160 --forward-only > $O/profile_config5.log 2>&1
cp gpurun_out/r4_prof_headline/pmc_traffic.json profiles/pmc_traffic.json; cp gpurun_out/r4_prof_config5/pmc_traffic.json profiles/pmc_traffic_config5.json
python bench.py > $O/bench_final.json 2> $O/bench_final.err
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
python bench.py --gaussians 500000 --no-camera-sequence > $O/bench_config2_500k.json 2> $O/bench_config2.err
python bench.py --gaussians 10000000 --width 3840 --height 2160 --forward-only --steps 100 --warmup 10 > $O/bench_config5_10M_4K_forward.json 2> $O/bench_config5.err
python bench.py --gaussians 3000000 --width 1600 --height 1200 --colors precomp --no-cpu-baseline --no-camera-sequence --steps 200 --warmup 20 > $O/bench_3M_1600x1200.json 2> $O/bench_3M.err
for m in 2 3 4; do python bench.py --scale-mult $m --no-cpu-baseline --no-camera-sequence --steps 200 --warmup 20 > $O/bench_dense_x$m.json 2> $O/bench_dense_x$m.err; done
python bench.py --no-cpu-baseline --no-camera-sequence --steps 300 --warmup 30 --option deterministic_backward=1 > $O/bench_deterministic.json 2> $O/bench_det.err
python bench.py --no-cpu-baseline --no-camera-sequence --steps 300 --warmup 30 --option exact_compositing=0 > $O/bench_exact_off.json 2> $O/bench_exact_off.err
timeout 600 python scripts/bench_wildgaussians_step.py --real-caller --steps 10 --warmup 3 > $O/real_caller_3M_plain.json 2> $O/rc1.err
timeout 600 python scripts/bench_wildgaussians_step.py --real-caller --steps 10 --warmup 3 --optins > $O/real_caller_3M_optins.json 2> $O/rc2.err
timeout 600 python scripts/bench_wildgaussians_step.py --real-caller --steps 10 --warmup 3 --optins --dual > $O/real_caller_3M_optins_two_colour_replay.json 2> $O/rc3.err
python tests/tools/stress_sweep_vs_reference.py 200000 20000 > $O/sweep_20k.txt 2>&1; tail -2 $O/sweep_20k.txt | tee -a $O/summary.txt
python - $O <<'PY' | tee -a $O/summary.txt
import json,sys,glob
for f in sorted(glob.glob(sys.argv[1]+"/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); s=d.get("stages_ms",{})
        if "value" in d:
            print(f"{f.split('/')[-1]:44s} {d['value']:8.1f} {d['unit']} fwd {d.get('forward_fps',0):8.1f} fps ms/step {d['ms_per_step']} q {d.get('step_ms_quantiles')} stages {s}")
            r=d.get("roofline"); 
            if r: print("      roofline", {k:r.get(k) for k in ("bound","kernel","frac","valu_busy","avg_launch_ms")}, "8d", r.get("hbm",{}).get("by_survey_8d_bytes"), "traffic", r.get("traffic"))
            for k in ("parity","cpu_baseline","camera_sequence","speedup_vs_reference_on_this_gpu"):
                if k in d: print("     ",k,d[k])
            if "reference_on_this_gpu" in d: print("      ref", {k:v for k,v in d["reference_on_this_gpu"].items() if k!="what"})
        else:
            print(f.split('/')[-1], {k:v for k,v in d.items() if 'ms' in k or 'share' in k})
    except Exception as e: print(f, "FAILED", e)
PY
