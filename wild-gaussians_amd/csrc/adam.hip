// Fused Adam step over a list of parameter tensors (include/wg_adam.h).  Reference semantics: the optimizer the
// training loop steps at wildgaussians/method.py:2019, built at :1030-1049 (torch.optim.Adam, one group per Gaussian attribute).
// A streaming kernel: 16 B read and 12 B written per element, nothing else.  One launch covers up to WG_ADAM_MAX_TENSORS tensors:
// the descriptors travel as a kernel argument, workgroup b finds its tensor in a prefix table of workgroup counts, and every
// thread moves four consecutive elements with 16-byte accesses (torch tensors start 256-byte aligned; the ragged tail of a
// tensor is done element by element).
#include <hip/hip_runtime.h>
#include "wg_adam.h"
#include "wg_rasterizer.h"

namespace wg {

constexpr int ADAM_THREADS = 256;
constexpr int ADAM_PER_THREAD = 4;
constexpr int ADAM_PER_BLOCK = ADAM_THREADS * ADAM_PER_THREAD * 4;  // four float4 per thread in flight

struct AdamBatch {
    wg_adam_tensor t[WG_ADAM_MAX_TENSORS];
    uint32_t first_block[WG_ADAM_MAX_TENSORS + 1];  // exclusive prefix of the tensors' workgroup counts
    int n;
};

typedef float f4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, const wg_adam_tensor& d) {
    g = d.weight_decay != 0.0f ? g + d.weight_decay * p : g;
    m = m + d.one_minus_beta1 * (g - m);          // exp_avg.lerp_(grad, 1 - beta1)   (weight < 0.5: start + weight * (end - start))
    v = d.beta2 * v + d.one_minus_beta2 * (g * g);  // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
    const float denom = sqrtf(v) / d.bias_correction2_sqrt + d.eps;
    p = p - d.step_size * (m / denom);            // param.addcdiv_(exp_avg, denom, value=-step_size)
}

__global__ void __launch_bounds__(ADAM_THREADS) fused_adam_kernel(const AdamBatch batch) {
    // the workgroup's tensor: a short linear search over wave-uniform values (SGPRs)
    int k = 0;
    while (k + 1 < batch.n && blockIdx.x >= batch.first_block[k + 1]) k++;
    const wg_adam_tensor& d = batch.t[k];
    const size_t base = (size_t)(blockIdx.x - batch.first_block[k]) * ADAM_PER_BLOCK;
    const bool vec = ((reinterpret_cast<uintptr_t>(d.param) | reinterpret_cast<uintptr_t>(d.grad) | reinterpret_cast<uintptr_t>(d.exp_avg) |
                       reinterpret_cast<uintptr_t>(d.exp_avg_sq)) & 15u) == 0;
    // a workgroup wholly inside its tensor (all but the last of each): every thread requests its sixteen 16-byte loads before it
    // computes or stores anything -- written as "load, update, store" per float4 the stores of one would sit between the loads of
    // the next (the four arrays may alias as far as the compiler knows), with four loads in flight per thread instead of sixteen
    if (vec && base + ADAM_PER_BLOCK <= d.numel) {
        float* __restrict__ const P = d.param;
        const float* __restrict__ const G = d.grad;
        float* __restrict__ const M = d.exp_avg;
        float* __restrict__ const V = d.exp_avg_sq;
        float4 p[4], g[4], m[4], v[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const size_t i = base + ((size_t)r * ADAM_THREADS + threadIdx.x) * ADAM_PER_THREAD;
            p[r] = *reinterpret_cast<const float4*>(P + i);
            const f4v gv = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(G + i));   // read once, by nobody else
            g[r] = make_float4(gv.x, gv.y, gv.z, gv.w);
            m[r] = *reinterpret_cast<const float4*>(M + i);
            v[r] = *reinterpret_cast<const float4*>(V + i);
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {
            adam_one(p[r].x, g[r].x, m[r].x, v[r].x, d);
            adam_one(p[r].y, g[r].y, m[r].y, v[r].y, d);
            adam_one(p[r].z, g[r].z, m[r].z, v[r].z, d);
            adam_one(p[r].w, g[r].w, m[r].w, v[r].w, d);
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const size_t i = base + ((size_t)r * ADAM_THREADS + threadIdx.x) * ADAM_PER_THREAD;
            *reinterpret_cast<float4*>(P + i) = p[r];
            *reinterpret_cast<float4*>(M + i) = m[r];
            *reinterpret_cast<float4*>(V + i) = v[r];
        }
        return;
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const size_t i = base + ((size_t)r * ADAM_THREADS + threadIdx.x) * ADAM_PER_THREAD;
        if (i >= d.numel) break;
        if (vec && i + ADAM_PER_THREAD <= d.numel) {
            float4 p = *reinterpret_cast<const float4*>(d.param + i);
            const float4 g = *reinterpret_cast<const float4*>(d.grad + i);
            float4 m = *reinterpret_cast<const float4*>(d.exp_avg + i);
            float4 v = *reinterpret_cast<const float4*>(d.exp_avg_sq + i);
            adam_one(p.x, g.x, m.x, v.x, d);
            adam_one(p.y, g.y, m.y, v.y, d);
            adam_one(p.z, g.z, m.z, v.z, d);
            adam_one(p.w, g.w, m.w, v.w, d);
            *reinterpret_cast<float4*>(d.param + i) = p;
            *reinterpret_cast<float4*>(d.exp_avg + i) = m;
            *reinterpret_cast<float4*>(d.exp_avg_sq + i) = v;
        } else {
            for (size_t j = i; j < d.numel && j < i + ADAM_PER_THREAD; j++) {
                float p = d.param[j], m = d.exp_avg[j], v = d.exp_avg_sq[j];
                adam_one(p, d.grad[j], m, v, d);
                d.param[j] = p;
                d.exp_avg[j] = m;
                d.exp_avg_sq[j] = v;
            }
        }
    }
}

}  // namespace wg

extern "C" int wg_fused_adam(int n_tensors, const wg_adam_tensor* tensors, void* stream) {
    if (n_tensors < 0 || (n_tensors > 0 && tensors == nullptr)) return WG_ERR_INVALID_ARGUMENT;
    for (int i = 0; i < n_tensors; i++) {
        const wg_adam_tensor& d = tensors[i];
        if (d.numel == 0) continue;
        if (!d.param || !d.grad || !d.exp_avg || !d.exp_avg_sq) return WG_ERR_INVALID_ARGUMENT;
        if (!(d.bias_correction2_sqrt > 0.0f) || !(d.one_minus_beta1 < 0.5f)) return WG_ERR_INVALID_ARGUMENT;  // step >= 1; torch's lerp changes formula at weight 0.5
        if (d.numel > (size_t)wg::ADAM_PER_BLOCK * 0x7fffffffull) return WG_ERR_INVALID_ARGUMENT;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    int i = 0;
    while (i < n_tensors) {
        wg::AdamBatch b;
        b.n = 0;
        uint64_t blocks = 0;
        while (i < n_tensors && b.n < WG_ADAM_MAX_TENSORS) {
            const wg_adam_tensor& d = tensors[i];
            const uint64_t nb = (d.numel + wg::ADAM_PER_BLOCK - 1) / wg::ADAM_PER_BLOCK;
            if (nb == 0) { i++; continue; }
            if (blocks + nb > 0x7fffffffull) break;  // grid limit: the rest goes into the next launch
            b.t[b.n] = d;
            b.first_block[b.n] = (uint32_t)blocks;
            blocks += nb;
            b.n++;
            i++;
        }
        if (b.n == 0) {
            if (i < n_tensors) return WG_ERR_INVALID_ARGUMENT;  // a single tensor beyond the grid limit (excluded above)
            break;
        }
        b.first_block[b.n] = (uint32_t)blocks;
        hipLaunchKernelGGL(wg::fused_adam_kernel, dim3((uint32_t)blocks), dim3(wg::ADAM_THREADS), 0, s, b);
        if (hipGetLastError() != hipSuccess) return WG_ERR_HIP;
    }
    return WG_OK;
}
