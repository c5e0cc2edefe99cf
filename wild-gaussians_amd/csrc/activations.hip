// Fused Gaussian activations + 3-D filter, forward and backward (include/wg_activations.h).
// Reference semantics: wildgaussians/method.py:1060-1086 with scaling_activation = exp, opacity_activation = sigmoid,
// rotation_activation = F.normalize.  Streaming kernels, one Gaussian per lane: 36 B in / 32 B out forward, 68 B in / 32 B
// out backward; everything is recomputed from the raw parameters in the backward pass (nothing is saved).
#include <hip/hip_runtime.h>
#include "wg_activations.h"
#include "wg_rasterizer.h"
#include "wg_act.h"

namespace wg {

__global__ void __launch_bounds__(256) activations_forward_kernel(int P, const float4* __restrict__ raw_rot, const float* __restrict__ raw_scale,
                                                                  const float* __restrict__ raw_opac, const float* __restrict__ filter,
                                                                  float4* __restrict__ rot, float* __restrict__ scale, float* __restrict__ opac) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const ActFwd a = act_forward(raw_rot[i], raw_scale[3 * i], raw_scale[3 * i + 1], raw_scale[3 * i + 2], raw_opac[i], filter[i]);
    rot[i] = a.q;
    scale[3 * i] = a.sc[0]; scale[3 * i + 1] = a.sc[1]; scale[3 * i + 2] = a.sc[2];
    opac[i] = a.o * a.coef;
}

__global__ void __launch_bounds__(256) activations_backward_kernel(int P, const float4* __restrict__ raw_rot, const float* __restrict__ raw_scale,
                                                                   const float* __restrict__ raw_opac, const float* __restrict__ filter,
                                                                   const float4* __restrict__ d_rot, const float* __restrict__ d_scale,
                                                                   const float* __restrict__ d_opac, float4* __restrict__ g_rot,
                                                                   float* __restrict__ g_scale, float* __restrict__ g_opac) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const float4 r = raw_rot[i];
    const ActFwd a = act_forward(r, raw_scale[3 * i], raw_scale[3 * i + 1], raw_scale[3 * i + 2], raw_opac[i], filter[i]);
    const float4 dq = d_rot ? d_rot[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float dsc[3] = {d_scale ? d_scale[3 * i] : 0.f, d_scale ? d_scale[3 * i + 1] : 0.f, d_scale ? d_scale[3 * i + 2] : 0.f};
    float4 gr;
    float gs[3], go;
    act_backward(a, dq, dsc, d_opac ? d_opac[i] : 0.f, gr, gs, go);
    g_rot[i] = gr;
    g_opac[i] = go;
    g_scale[3 * i] = gs[0]; g_scale[3 * i + 1] = gs[1]; g_scale[3 * i + 2] = gs[2];
}

}  // namespace wg

extern "C" {

int wg_activations_forward(int P, const float* raw_rotations, const float* raw_scales, const float* raw_opacities,
                           const float* filter_3D, float* rotations, float* scales, float* opacities, void* stream) {
    if (P < 0 || P > 0x7fffffff / 4) return WG_ERR_INVALID_ARGUMENT;  // per-Gaussian element indices (3 i, 4 i) are 32-bit
    if (P == 0) return WG_OK;
    if (!raw_rotations || !raw_scales || !raw_opacities || !filter_3D || !rotations || !scales || !opacities) return WG_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(wg::activations_forward_kernel, dim3((P + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), P,
                       reinterpret_cast<const float4*>(raw_rotations), raw_scales, raw_opacities, filter_3D,
                       reinterpret_cast<float4*>(rotations), scales, opacities);
    return hipGetLastError() == hipSuccess ? WG_OK : WG_ERR_HIP;
}

int wg_activations_backward(int P, const float* raw_rotations, const float* raw_scales, const float* raw_opacities,
                            const float* filter_3D, const float* dL_drotations, const float* dL_dscales, const float* dL_dopacities,
                            float* dL_draw_rotations, float* dL_draw_scales, float* dL_draw_opacities, void* stream) {
    if (P < 0 || P > 0x7fffffff / 4) return WG_ERR_INVALID_ARGUMENT;  // per-Gaussian element indices (3 i, 4 i) are 32-bit
    if (P == 0) return WG_OK;
    if (!raw_rotations || !raw_scales || !raw_opacities || !filter_3D || !dL_draw_rotations || !dL_draw_scales || !dL_draw_opacities)
        return WG_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(wg::activations_backward_kernel, dim3((P + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), P,
                       reinterpret_cast<const float4*>(raw_rotations), raw_scales, raw_opacities, filter_3D,
                       reinterpret_cast<const float4*>(dL_drotations), dL_dscales, dL_dopacities,
                       reinterpret_cast<float4*>(dL_draw_rotations), dL_draw_scales, dL_draw_opacities);
    return hipGetLastError() == hipSuccess ? WG_OK : WG_ERR_HIP;
}

}  // extern "C"
