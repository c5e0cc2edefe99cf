// Fused densification statistics (include/wg_densify.h).  Reference semantics:
// wildgaussians/method.py:1995-1998 and :1470-1477.  One Gaussian per lane; invisible Gaussians (radii <= 0) touch nothing
// but their radius, so a sparse view reads 4 B per Gaussian and the read-modify-write traffic scales with the visible count.
#include <hip/hip_runtime.h>
#include "wg_densify.h"
#include "wg_rasterizer.h"

namespace wg {

template <bool ABS, bool RADII>
__global__ void __launch_bounds__(256) densification_stats_kernel(int P, const int* __restrict__ radii, const float* __restrict__ grad,
                                                                  float* __restrict__ xyz_grad, float* __restrict__ accum_abs,
                                                                  float* __restrict__ accum_abs_max, float* __restrict__ denom,
                                                                  float* __restrict__ max_radii) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const int r = radii[i];
    if (r <= 0) return;
    const float gx = grad[3 * i], gy = grad[3 * i + 1];
    xyz_grad[i] += sqrtf(gx * gx + gy * gy);  // torch.norm(g[:, :2], dim=-1)
    if (ABS) {
        const float n = fabsf(grad[3 * i + 2]);  // the norm of a single element
        accum_abs[i] += n;
        accum_abs_max[i] = fmaxf(accum_abs_max[i], n);
    }
    denom[i] += 1.0f;
    if (RADII) max_radii[i] = fmaxf(max_radii[i], (float)r);  // torch.max(float, int32) promotes to float
}

}  // namespace wg

extern "C" int wg_densification_stats(int P, const int* radii, const float* viewspace_grad, float* xyz_grad, float* xyz_gradient_accum_abs,
                                      float* xyz_gradient_accum_abs_max, float* denom, float* max_radii2D, void* stream) {
    if (P < 0 || P > 0x7fffffff / 4) return WG_ERR_INVALID_ARGUMENT;  // per-Gaussian element indices (3 i, 4 i) are 32-bit
    if (P == 0) return WG_OK;
    if (!radii || !viewspace_grad || !xyz_grad || !denom) return WG_ERR_INVALID_ARGUMENT;
    if ((xyz_gradient_accum_abs == nullptr) != (xyz_gradient_accum_abs_max == nullptr)) return WG_ERR_INVALID_ARGUMENT;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const dim3 grid((P + 255) / 256), block(256);
    const bool a = xyz_gradient_accum_abs != nullptr, m = max_radii2D != nullptr;
#define WG_LAUNCH(A, M) \
    wg::densification_stats_kernel<A, M><<<grid, block, 0, s>>>(P, radii, viewspace_grad, xyz_grad, xyz_gradient_accum_abs, \
                                                                xyz_gradient_accum_abs_max, denom, max_radii2D)
    if (a && m) WG_LAUNCH(true, true);
    else if (a) WG_LAUNCH(true, false);
    else if (m) WG_LAUNCH(false, true);
    else WG_LAUNCH(false, false);
#undef WG_LAUNCH
    return hipGetLastError() == hipSuccess ? WG_OK : WG_ERR_HIP;
}
