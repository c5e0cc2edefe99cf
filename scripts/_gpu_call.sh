cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2c21; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -4 > $O/pytest_gpu.log; cat $O/pytest_gpu.log
bash scripts/profile_gpu.sh r2c21/prof_headline > $O/prof_headline.log 2>&1
cp $O/prof_headline/pmc_traffic.json profiles/pmc_traffic.json
timeout 900 python bench.py > $O/bench_final.log 2>&1; tail -1 $O/bench_final.log | cut -c1-200
WORKLOAD="10000000 Gaussians, 3840x2160, sh" bash scripts/profile_gpu.sh r2c21/prof_config5 --gaussians 10000000 --width 3840 --height 2160 --forward-only > $O/prof_config5.log 2>&1
cp $O/prof_config5/pmc_traffic.json profiles/pmc_traffic_config5.json
timeout 600 python bench.py --no-cpu-baseline --steps 50 --warmup 5 --gaussians 10000000 --width 3840 --height 2160 --forward-only > $O/bench_config5.log 2>&1; tail -1 $O/bench_config5.log | cut -c1-200
python __graft_entry__.py smoke 2>&1 | tail -2
