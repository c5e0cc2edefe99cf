// Fused Gaussian activations + 3-D filter, forward and backward (include/wg_activations.h).
// Reference semantics: wildgaussians/method.py:1060-1086 with scaling_activation = exp, opacity_activation = sigmoid,
// rotation_activation = F.normalize.  Streaming kernels, one Gaussian per lane: 36 B in / 32 B out forward, 68 B in / 32 B
// out backward; everything is recomputed from the raw parameters in the backward pass (nothing is saved).
#include <hip/hip_runtime.h>
#include "wg_activations.h"
#include "wg_rasterizer.h"

namespace wg {

struct ActFwd {
    float4 q;        // normalised rotation
    float inv_n;     // 1 / max(|raw rotation|, eps)
    bool clamped;    // the norm was below eps (F.normalize then divides by the constant eps)
    float rs[3];     // exp(raw scale)
    float sa[3];     // rs^2 + f^2
    float sc[3];     // sqrt(sa)
    float o, coef, det1, det2;
};

__device__ __forceinline__ ActFwd act_forward(float4 r, float s0, float s1, float s2, float ol, float f) {
    ActFwd a;
    const float n = sqrtf(r.x * r.x + r.y * r.y + r.z * r.z + r.w * r.w);
    a.clamped = n < 1e-12f;
    a.inv_n = 1.0f / fmaxf(n, 1e-12f);
    a.q = make_float4(r.x * a.inv_n, r.y * a.inv_n, r.z * a.inv_n, r.w * a.inv_n);
    const float raw[3] = {s0, s1, s2};
    const float f2 = f * f;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        a.rs[i] = expf(raw[i]);
        a.sa[i] = a.rs[i] * a.rs[i] + f2;
        a.sc[i] = sqrtf(a.sa[i]);
    }
    a.o = 1.0f / (1.0f + expf(-ol));
    a.det1 = (a.rs[0] * a.rs[0]) * (a.rs[1] * a.rs[1]) * (a.rs[2] * a.rs[2]);
    a.det2 = a.sa[0] * a.sa[1] * a.sa[2];
    a.coef = sqrtf(a.det1 / a.det2);
    return a;
}

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
    // rotation: q = v / max(|v|, eps)
    const float4 dq = d_rot ? d_rot[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 gr;
    if (a.clamped) {
        gr = make_float4(dq.x * a.inv_n, dq.y * a.inv_n, dq.z * a.inv_n, dq.w * a.inv_n);
    } else {
        const float dot = a.q.x * dq.x + a.q.y * dq.y + a.q.z * dq.z + a.q.w * dq.w;
        gr = make_float4((dq.x - a.q.x * dot) * a.inv_n, (dq.y - a.q.y * dot) * a.inv_n, (dq.z - a.q.z * dot) * a.inv_n,
                         (dq.w - a.q.w * dot) * a.inv_n);
    }
    g_rot[i] = gr;
    // opacity = sigmoid(ol) * coef,  coef = sqrt(det1 / det2)
    const float dop = d_opac ? d_opac[i] : 0.f;
    g_opac[i] = dop * a.coef * a.o * (1.0f - a.o);
    const float d_coef = dop * a.o;
    const float d_r = d_coef * 0.5f / a.coef;
    const float d_det1 = d_r / a.det2;
    const float d_det2 = -d_r * a.det1 / (a.det2 * a.det2);
    const float s2[3] = {a.rs[0] * a.rs[0], a.rs[1] * a.rs[1], a.rs[2] * a.rs[2]};
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int k1 = (k + 1) % 3, k2 = (k + 2) % 3;
        const float dsc = d_scale ? d_scale[3 * i + k] : 0.f;
        const float d_sa = d_det2 * (a.sa[k1] * a.sa[k2]) + dsc * 0.5f / a.sc[k];  // via det2 and via scales = sqrt(sa)
        const float d_s2 = d_det1 * (s2[k1] * s2[k2]) + d_sa;                       // via det1 and via sa = s2 + f^2
        g_scale[3 * i + k] = d_s2 * 2.0f * s2[k];                                   // s2 = exp(raw)^2: d s2 / d raw = 2 s2
    }
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
