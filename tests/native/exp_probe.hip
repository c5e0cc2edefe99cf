// tests/native/exp_probe.hip -- TEST INFRASTRUCTURE: holds wg::ref_expf_nonpos (csrc/wg_alpha.h: the hand-restated float32 `exp`
// expansion the decision-exact compositing rests on) to THE COMPILER'S OWN lowering of `exp(float)` -- what the reference's
// `exp(power)` (forward.cu:364, backward.cu:543) compiles to with this toolchain -- over EVERY float32 argument in [-104, -0]: 1.12e9 values.
// Built with -ffp-contract=off like oracle/_ref's no-contraction build (wild-gaussians_amd/build.py: build_exp_probe).  A ROCm upgrade
// that changes the exp lowering turns tests/test_exp_expansion.py red directly, not only the parity tests downstream.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "wg_alpha.h"

namespace {
// out[0] = arguments in [-87, -0] whose two results differ in any bit, out[1] = the first such argument's bits (min),
// out[2] = arguments in [-104, -87) where either result is not below 1e-30 (both are then far below any alpha >= 1/255), out[3] = arguments tested
__global__ void __launch_bounds__(256) exp_probe_kernel(uint32_t last_bits, unsigned long long* out) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long bad = 0, tiny_bad = 0, n = 0, first = ~0ull;
    for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b <= last_bits; b += stride) {
        const float x = -__uint_as_float((uint32_t)b);
        const float want = exp(x);                    // the reference's call, as this compiler lowers it
        const float got = wg::ref_expf_nonpos(x);
        n++;
        if (x >= -87.0f) {
            if (__float_as_uint(want) != __float_as_uint(got)) { bad++; if (b < first) first = b; }
        } else if (!(want < 1e-30f) || !(got < 1e-30f)) {
            tiny_bad++;
        }
    }
    atomicAdd(&out[0], bad);
    atomicMin(&out[1], first);
    atomicAdd(&out[2], tiny_bad);
    atomicAdd(&out[3], n);
}
}  // namespace

extern "C" int exp_probe_run(unsigned long long* host_out4) {
    unsigned long long* d = nullptr;
    if (hipMalloc(&d, 4 * sizeof(unsigned long long)) != hipSuccess) return -1;
    const unsigned long long init[4] = {0ull, ~0ull, 0ull, 0ull};
    if (hipMemcpy(d, init, sizeof(init), hipMemcpyHostToDevice) != hipSuccess) return -2;
    union { float f; uint32_t u; } last;
    last.f = 104.0f;
    hipLaunchKernelGGL(exp_probe_kernel, dim3(256 * 64), dim3(256), 0, 0, last.u, d);
    if (hipDeviceSynchronize() != hipSuccess) return -3;
    if (hipMemcpy(host_out4, d, sizeof(init), hipMemcpyDeviceToHost) != hipSuccess) return -4;
    (void)hipFree(d);
    return 0;
}
