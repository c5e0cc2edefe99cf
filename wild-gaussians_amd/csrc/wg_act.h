// Gaussian activations + 3-D filter as device functions (reference semantics: wildgaussians/method.py:1060-1086 with scaling_activation =
// exp, opacity_activation = sigmoid, rotation_activation = F.normalize), shared by the stand-alone kernels (activations.hip) and by
// the preprocess kernels' raw-parameter mode (wg_raw_gaussians in include/wg_rasterizer.h).
#pragma once
#include <hip/hip_runtime.h>

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

// No fused multiply-adds in either function, whatever the flags of the file that includes them (the pragma binds to the function body):
// the preprocess kernel (built -ffp-contract=off), the stand-alone activation kernels (likewise) and the preprocess-BACKWARD kernel
// (built with the compiler's default contraction) all recompute act_forward() and must get the same bits -- the backward pass chains
// through the very q / scales / coef the forward frame was made of.
__device__ __forceinline__ ActFwd act_forward(float4 r, float s0, float s1, float s2, float ol, float f) {
#pragma clang fp contract(off)
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

// gradients w.r.t. the raw parameters from those w.r.t. the activated ones (dq: rotation, dsc: filtered scales, dop: filtered opacity)
__device__ __forceinline__ void act_backward(const ActFwd& a, float4 dq, const float dsc[3], float dop, float4& g_rot, float g_scale[3],
                                             float& g_opac) {
#pragma clang fp contract(off)
    // rotation: q = v / max(|v|, eps)
    if (a.clamped) {
        g_rot = make_float4(dq.x * a.inv_n, dq.y * a.inv_n, dq.z * a.inv_n, dq.w * a.inv_n);
    } else {
        const float dot = a.q.x * dq.x + a.q.y * dq.y + a.q.z * dq.z + a.q.w * dq.w;
        g_rot = make_float4((dq.x - a.q.x * dot) * a.inv_n, (dq.y - a.q.y * dot) * a.inv_n, (dq.z - a.q.z * dot) * a.inv_n,
                            (dq.w - a.q.w * dot) * a.inv_n);
    }
    // opacity = sigmoid(ol) * coef,  coef = sqrt(det1 / det2)
    g_opac = dop * a.coef * a.o * (1.0f - a.o);
    const float d_coef = dop * a.o;
    const float d_r = d_coef * 0.5f / a.coef;
    const float d_det1 = d_r / a.det2;
    const float d_det2 = -d_r * a.det1 / (a.det2 * a.det2);
    const float s2[3] = {a.rs[0] * a.rs[0], a.rs[1] * a.rs[1], a.rs[2] * a.rs[2]};
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int k1 = (k + 1) % 3, k2 = (k + 2) % 3;
        const float d_sa = d_det2 * (a.sa[k1] * a.sa[k2]) + dsc[k] * 0.5f / a.sc[k];  // via det2 and via scales = sqrt(sa)
        const float d_s2 = d_det1 * (s2[k1] * s2[k2]) + d_sa;                          // via det1 and via sa = s2 + f^2
        g_scale[k] = d_s2 * 2.0f * s2[k];                                              // s2 = exp(raw)^2: d s2 / d raw = 2 s2
    }
}

}  // namespace wg
