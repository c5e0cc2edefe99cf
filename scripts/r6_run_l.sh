O=gpurun_out/r6l; mkdir -p $O
B=wild-gaussians_amd/build/malloc_async_lost_stores
for a in "40 320 0" "40 320 3" "40 320 0 malloc" "40 320 3 malloc" "60 64 3" "30 1024 2"; do echo "== $a"; timeout 300 $B $a 2>&1 | tail -4; echo "rc=$?"; done | tee $O/malloc_async_lost_stores.txt
(time python -m pytest tests -m gpu -x -q) > $O/pytest_gpu.log 2>&1; grep -n "passed\|failed" $O/pytest_gpu.log | tail -2
