cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2c5; mkdir -p $O
timeout 600 python -m pytest tests/test_ssim.py -m gpu -q --tb=short 2>&1 | tail -40 > $O/pytest_ssim.log; tail -30 $O/pytest_ssim.log
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -6 > $O/pytest_gpu.log; cat $O/pytest_gpu.log
timeout 900 scripts/asan_pass.sh $O/asan_pass.log; tail -12 $O/asan_pass.log
WG_ROCTX=1 timeout 600 rocprofv3 --marker-trace --kernel-trace --stats -d $O/mk -o m -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-profile > $O/mk.log 2>&1
python - <<'PY' > gpurun_out/r2c5/roctx_marker_summary.txt 2>&1
import sqlite3, glob
db = sqlite3.connect(glob.glob("gpurun_out/r2c5/mk/*results.db")[0])
print(list(db.execute("select count(*) from regions")))
for r in db.execute("select category, name, count(*), avg(duration)/1e3 from regions group by category, name order by count(*) desc limit 40"):
    print(r)
PY
cat $O/roctx_marker_summary.txt | head -45; ls $O/mk; rm -rf $O/mk
timeout 300 python bench.py --steps 200 --warmup 30 --no-cpu-baseline | tail -1 | cut -c1-400
