# the torch-free driver under ASan, small scene, three times with asan_pass.sh's options: is the sanitizer's exit-time CHECK (inside libhsa's static destructors) behind the driver's "ok"?
cd "$GRAFT_REPO_ROOT"
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:abort_on_error=0:protect_shadow_gap=0
for i in 1 2 3; do
  wild-gaussians_amd/build/asan/c_abi_driver > gpurun_out/asan_driver_small_$i.log 2>&1; echo "run $i rc=$?"
  grep -n "^ok \|CHECK failed\|ERROR: AddressSanitizer\|SUMMARY\|cxa_finalize" gpurun_out/asan_driver_small_$i.log | head -6
done
