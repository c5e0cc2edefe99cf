#!/bin/bash
# round 6, final measurements on the round's final sources (run on the GPU box: gpurun -- 'bash scripts/r6/r6_final.sh'; the butterfly variant build of K9 must exist:
# WG_BUILD_VARIANT=butterfly WG_EXTRA_FLAGS=-DWG_BWD_LDS_REDUCE=0 python wild-gaussians_amd/build.py): profiles with the
# lane-utilisation counters, pair counts from the counting build, the bench line of every BASELINE config, config 4's N = 1 point, A/Bs.
O=gpurun_out/r6final; mkdir -p $O
(time python -m pytest tests -m gpu -q) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -4 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
bash scripts/r6/pmc_calibrate.sh r6final_cal > $O/pmc_calibrate.log 2>&1
wild-gaussians_amd/build/malloc_async_lost_stores 40 320 0 > $O/malloc_async_lost_stores.txt 2>&1; wild-gaussians_amd/build/malloc_async_lost_stores 40 320 0 malloc >> $O/malloc_async_lost_stores.txt 2>&1
bash scripts/profile_gpu.sh r6_prof_headline > $O/profile_headline.log 2>&1
WORKLOAD="10000000 Gaussians, 3840x2160, sh" bash scripts/profile_gpu.sh r6_prof_config5 --gaussians 10000000 --width 3840 --height 2160 --forward-only > $O/profile_config5.log 2>&1
cp gpurun_out/r6_prof_headline/pmc_traffic.json profiles/pmc_traffic.json; cp gpurun_out/r6_prof_config5/pmc_traffic.json profiles/pmc_traffic_config5.json
python tests/tools/count_pairs.py > $O/pair_counts.json 2> $O/pair_counts.err
python tests/tools/count_pairs.py --gaussians 10000000 --width 3840 --height 2160 --forward-only --oracle-gaussians 2000000 > $O/pair_counts_config5.json 2> $O/pair_counts_c5.err
cp $O/pair_counts.json profiles/pair_counts.json; cp $O/pair_counts_config5.json profiles/pair_counts_config5.json
python bench.py > $O/bench_final.json 2> $O/bench_final.err
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
python bench.py --gpus 1 --views 8 --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_config4_views8_gpus1.json 2> $O/bench_views8.err
for st in 2 3; do python bench.py --gpus 1 --views 8 --streams $st --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_config4_views8_gpus1_streams$st.json 2> $O/bench_views8_s$st.err; done
python bench.py --no-cpu-baseline --no-camera-sequence --steps 300 --warmup 30 --option forward_order=0 > $O/bench_forward_order_off.json 2> $O/bench_fo_off.err
python bench.py --gaussians 10000000 --width 3840 --height 2160 --forward-only --steps 100 --warmup 10 --no-cpu-baseline --option near_adapt=0 --option forward_order=0 > $O/bench_config5_round5_settings.json 2> $O/bench_config5_r5.err
python scripts/probe_balance.py > $O/probe_balance_headline.json 2> $O/probe.err
python scripts/probe_balance.py --option forward_order=0 > $O/probe_balance_headline_forward_order_off.json 2>> $O/probe.err
(cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && rocprofv3 --kernel-trace -d $O/trace_s2 -o t -- python bench.py --views 8 --gpus 1 --streams 2 --steps 10 --warmup 3 --no-cpu-baseline --no-profile > $O/trace_s2.log 2>&1; python scripts/stream_overlap.py $O/trace_s2/t_results.db > $O/stream_overlap_views8_streams2.txt 2>&1; rm -rf $O/trace_s2)
python bench.py --gaussians 500000 --no-camera-sequence > $O/bench_config2_500k.json 2> $O/bench_config2.err
python bench.py --gaussians 10000000 --width 3840 --height 2160 --forward-only --steps 100 --warmup 10 > $O/bench_config5_10M_4K_forward.json 2> $O/bench_config5.err
python bench.py --gaussians 3000000 --width 1600 --height 1200 --colors precomp --no-cpu-baseline --no-camera-sequence --steps 200 --warmup 20 > $O/bench_3M_1600x1200.json 2> $O/bench_3M.err
for m in 2 3 4; do python bench.py --scale-mult $m --no-cpu-baseline --no-camera-sequence --steps 200 --warmup 20 > $O/bench_dense_x$m.json 2> $O/bench_dense_x$m.err; done
python bench.py --no-cpu-baseline --no-camera-sequence --steps 300 --warmup 30 --option sh_stream=0 > $O/bench_sh_stream_off.json 2> $O/bench_sh_stream_off.err
python bench.py --gaussians 10000000 --width 3840 --height 2160 --forward-only --no-cpu-baseline --no-camera-sequence --steps 1500 --warmup 20 > $O/bench_config5_1500_steps.json 2> $O/bench_config5_1500.err
python scripts/r6/near_trace.py 2500 > $O/near_trace_synchronized.txt 2> $O/near_trace_sync.err
NEAR_TRACE_BENCH_LIKE=1 python scripts/r6/near_trace.py 3500 > $O/near_trace_pipelined.txt 2> $O/near_trace_pipe.err
python scripts/r6/diag_sh_stream.py > $O/diag_sh_stream_1M.json 2> $O/diag_sh_stream.err
WG_RASTERIZER_LIB=$PWD/wild-gaussians_amd/build/butterfly/libwg_rasterizer.so python bench.py --no-cpu-baseline --no-camera-sequence --no-config-legs --steps 300 --warmup 30 > $O/bench_k9_butterfly_build.json 2> $O/bench_k9_butterfly.err
python bench.py --no-cpu-baseline --no-camera-sequence --steps 300 --warmup 30 --option deterministic_backward=1 > $O/bench_deterministic.json 2> $O/bench_det.err
python bench.py --no-cpu-baseline --no-camera-sequence --steps 300 --warmup 30 --option exact_compositing=0 > $O/bench_exact_off.json 2> $O/bench_exact_off.err
python scripts/diag_host_wait.py > $O/host_wait_1M_compiled_binding.json 2> $O/hw1.err
WG_BINDING=ctypes python scripts/diag_host_wait.py > $O/host_wait_1M_ctypes_binding.json 2> $O/hw2.err
timeout 600 python scripts/bench_wildgaussians_step.py --real-caller --steps 10 --warmup 3 > $O/real_caller_3M_plain.json 2> $O/rc1.err
timeout 600 python scripts/bench_wildgaussians_step.py --real-caller --steps 10 --warmup 3 --optins --two-tone-edit > $O/real_caller_3M_optins_two_tone.json 2> $O/rc2.err
python - $O <<'PY' | tee $O/summary.txt
import json,sys,glob
for f in sorted(glob.glob(sys.argv[1]+"/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); s=d.get("stages_ms",{})
        if "value" in d:
            print(f"{f.split('/')[-1]:44s} {d['value']:8.1f} {d['unit']} fwd {d.get('forward_fps',0):8.1f} fps ms/step {d['ms_per_step']} q {d.get('step_ms_quantiles')} stages {s}")
            r=d.get("roofline")
            if r: print("      roofline", {k:r.get(k) for k in ("bound","kernel","achieved","frac","avg_launch_ms","traffic","traffic_over_algorithmic_bytes")}, "compute", {k:v for k,v in (r.get("compute") or {}).items() if k in ("useful_TFLOPs","frac_of_fp32_plain_peak","frac_of_fp32_packed_peak","valu")})
            for k in ("parity","cpu_baseline","render_compute","region_with_cycle_collector_on","per_rank_num_rendered","speedup_vs_reference_on_this_gpu"):
                if k in d: print("     ",k,json.dumps(d[k])[:700])
            if "camera_sequence" in d: print("      camera_sequence.steady", d["camera_sequence"]["steady"])
        else:
            print(f.split('/')[-1], json.dumps(d)[:600])
    except Exception as e: print(f, "FAILED", e)
PY
