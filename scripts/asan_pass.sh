#!/bin/bash
# AddressSanitizer pass over the HOST side of libwg_rasterizer.so ("build adds"): the C-ABI glue, scratch
# carving, option / profiler state, allocator callbacks.  Device code is not instrumented (-fno-gpu-sanitize).
# Builds wild-gaussians_amd/build/asan/libwg_rasterizer.so, then runs the host tests and -- when a GPU is present -- a slice of
# the GPU parity tests through it.  usage: scripts/asan_pass.sh [log file]
set -u
cd "$(dirname "$0")/.."
LOG=${1:-/dev/stdout}
WG_BUILD_VARIANT=asan WG_EXTRA_FLAGS="-fsanitize=address -fno-gpu-sanitize -fno-omit-frame-pointer -g" python wild-gaussians_amd/build.py --driver > /dev/null || exit 1
export LD_PRELOAD=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
# protect_shadow_gap=0: the HIP runtime reserves address space inside ASan's shadow gap (as CUDA does)
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:abort_on_error=0:protect_shadow_gap=0
export WG_RASTERIZER_LIB=$PWD/wild-gaussians_amd/build/asan/libwg_rasterizer.so
{
  # 1. no interpreter in between: the torch-free driver (tests/native/c_abi_driver.cpp), itself built with ASan, runs a forward +
  #    backward + markVisible through the instrumented library on the GPU (nothing to run without one)
  if [ -e /dev/kfd ]; then
    echo "== C-ABI driver under ASan (forward + backward + markVisible + recolor + two-colour + raw-parameter + two-tone calls, three deterministic passes over one frame, three concurrent callers; 20000 Gaussians @ 320x200, then 200000 @ 1280x720)"
    env -u LD_PRELOAD wild-gaussians_amd/build/asan/c_abi_driver 2>&1 | tail -20; echo "rc=${PIPESTATUS[0]}"
    env -u LD_PRELOAD wild-gaussians_amd/build/asan/c_abi_driver 200000 1280 720 2>&1 | tail -20; echo "rc=${PIPESTATUS[0]}"
  fi
  echo "== host tests under ASan ($WG_RASTERIZER_LIB)"
  python -m pytest tests/test_host_cpu.py -q -k "export or invalid or scratch or options_round" 2>&1 | tail -15
  echo "rc=${PIPESTATUS[0]}"
  if python -c "import torch,sys; sys.exit(0 if torch.cuda.is_available() else 1)" 2>/dev/null; then
    echo "== GPU slice under ASan"
    python -m pytest tests/test_parity_gpu.py -q -k "operator_surface or all_culled or c_abi_backward or gradient_record or two_tones or (sweep_against and (3 or 7 or 11))" 2>&1 | tail -15
    echo "rc=${PIPESTATUS[0]}"
    python -m pytest tests/test_knn.py tests/test_ssim.py tests/test_densify.py tests/test_activations.py -q -m gpu 2>&1 | tail -5
    echo "rc=${PIPESTATUS[0]}"
  else
    echo "== no GPU visible to torch under ASan (rc of the probe: $?)"
    python -c "import torch; print(torch.cuda.is_available())" 2>&1 | tail -5
  fi
} > "$LOG" 2>&1
grep -c "ERROR: AddressSanitizer" "$LOG" | sed 's/^/asan errors: /'
