// Fused eval_sh (include/wg_sh_eval.h).  Reference semantics: wildgaussians/method.py:493-548 (constants :461-479),
// called at :1555-1565 and :1596-1598.  One Gaussian per lane, both ways a streaming kernel: 12 K + 12 B in, 12 B out forward;
// 12 K + 24 B in, 12 K + 12 B out backward.
#include <hip/hip_runtime.h>
#include "wg_sh_eval.h"
#include "wg_rasterizer.h"

namespace wg {

constexpr float SHE_C0 = 0.28209479177387814f;
constexpr float SHE_C1 = 0.4886025119029199f;
constexpr float SHE_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f, 0.5462742152960396f};
constexpr float SHE_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                             -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f};

// the (deg + 1)^2 basis values at a direction
template <int DEG>
__device__ __forceinline__ void sh_basis(float x, float y, float z, float* b) {
    b[0] = SHE_C0;
    if (DEG > 0) {
        b[1] = -SHE_C1 * y;
        b[2] = SHE_C1 * z;
        b[3] = -SHE_C1 * x;
    }
    if (DEG > 1) {
        const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
        b[4] = SHE_C2[0] * xy;
        b[5] = SHE_C2[1] * yz;
        b[6] = SHE_C2[2] * (2.0f * zz - xx - yy);
        b[7] = SHE_C2[3] * xz;
        b[8] = SHE_C2[4] * (xx - yy);
        if (DEG > 2) {
            b[9] = SHE_C3[0] * y * (3.0f * xx - yy);
            b[10] = SHE_C3[1] * xy * z;
            b[11] = SHE_C3[2] * y * (4.0f * zz - xx - yy);
            b[12] = SHE_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy);
            b[13] = SHE_C3[4] * x * (4.0f * zz - xx - yy);
            b[14] = SHE_C3[5] * z * (xx - yy);
            b[15] = SHE_C3[6] * x * (xx - 3.0f * yy);
        }
    }
}

// a channel's coefficients: 16-byte loads when the row is 16-byte aligned (K % 4 == 0: always for the reference's K = 16)
template <int N>
__device__ __forceinline__ void load_row(const float* __restrict__ p, bool vec, float* v) {
    if (vec && N % 4 == 0) {
#pragma unroll
        for (int q = 0; q < N / 4; q++) {
            const float4 t = reinterpret_cast<const float4*>(p)[q];
            v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
        }
    } else {
#pragma unroll
        for (int k = 0; k < N; k++) v[k] = p[k];
    }
}

template <int DEG>
__global__ void __launch_bounds__(256) eval_sh_forward_kernel(int P, int K, const float* __restrict__ sh, const float* __restrict__ dirs,
                                                              float* __restrict__ out) {
    constexpr int N = (DEG + 1) * (DEG + 1);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    float b[N];
    const size_t i3 = 3 * (size_t)i;
    sh_basis<DEG>(dirs[i3], dirs[i3 + 1], dirs[i3 + 2], b);
    const bool vec = (K & 3) == 0 && (reinterpret_cast<uintptr_t>(sh) & 15u) == 0;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        float v[N];
        load_row<N>(sh + ((size_t)i * 3 + c) * K, vec, v);
        float r = b[0] * v[0];
#pragma unroll
        for (int k = 1; k < N; k++) r += b[k] * v[k];
        out[i3 + c] = r;
    }
}

template <int DEG>
__global__ void __launch_bounds__(256) eval_sh_backward_kernel(int P, int K, const float* __restrict__ sh, const float* __restrict__ dirs,
                                                               const float* __restrict__ grad_out, float* __restrict__ grad_sh,
                                                               float* __restrict__ grad_dirs) {
    constexpr int N = (DEG + 1) * (DEG + 1);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const size_t i3 = 3 * (size_t)i;
    const float x = dirs[i3], y = dirs[i3 + 1], z = dirs[i3 + 2];
    float b[N];
    sh_basis<DEG>(x, y, z, b);
    const float g[3] = {grad_out[i3], grad_out[i3 + 1], grad_out[i3 + 2]};
    const bool vec = (K & 3) == 0 && ((reinterpret_cast<uintptr_t>(sh) | reinterpret_cast<uintptr_t>(grad_sh)) & 15u) == 0;
    float s[N];  // s_k = sum_c grad_out[c] * sh[c][k]: what the direction's gradient needs of the coefficients
#pragma unroll
    for (int k = 0; k < N; k++) s[k] = 0.f;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        float* row = grad_sh + ((size_t)i * 3 + c) * K;
        if (grad_dirs && DEG > 0) {
            float v[N];
            load_row<N>(sh + ((size_t)i * 3 + c) * K, vec, v);
#pragma unroll
            for (int k = 0; k < N; k++) s[k] += g[c] * v[k];
        }
        // d out[c] / d sh[c][k] = basis_k; coefficients beyond the active degree take no gradient
        if (vec && K == 16) {  // the reference's layout: compile-time indices keep the basis in registers
#pragma unroll
            for (int q = 0; q < 4; q++) {
                float4 t;
                t.x = 4 * q < N ? b[4 * q < N ? 4 * q : 0] * g[c] : 0.f;
                t.y = 4 * q + 1 < N ? b[4 * q + 1 < N ? 4 * q + 1 : 0] * g[c] : 0.f;
                t.z = 4 * q + 2 < N ? b[4 * q + 2 < N ? 4 * q + 2 : 0] * g[c] : 0.f;
                t.w = 4 * q + 3 < N ? b[4 * q + 3 < N ? 4 * q + 3 : 0] * g[c] : 0.f;
                reinterpret_cast<float4*>(row)[q] = t;
            }
        } else if (vec) {
            for (int q = 0; q < K / 4; q++) {
                float4 t;
                t.x = 4 * q < N ? b[min(4 * q, N - 1)] * g[c] : 0.f;
                t.y = 4 * q + 1 < N ? b[min(4 * q + 1, N - 1)] * g[c] : 0.f;
                t.z = 4 * q + 2 < N ? b[min(4 * q + 2, N - 1)] * g[c] : 0.f;
                t.w = 4 * q + 3 < N ? b[min(4 * q + 3, N - 1)] * g[c] : 0.f;
                reinterpret_cast<float4*>(row)[q] = t;
            }
        } else {
            for (int k = 0; k < K; k++) row[k] = k < N ? b[min(k, N - 1)] * g[c] : 0.f;
        }
    }
    if (grad_dirs) {
        float gx = 0.f, gy = 0.f, gz = 0.f;
        if (DEG > 0) {
            gy += -SHE_C1 * s[1 % N];
            gz += SHE_C1 * s[2 % N];
            gx += -SHE_C1 * s[3 % N];
        }
        if (DEG > 1) {
            const float s4 = SHE_C2[0] * s[4 % N], s5 = SHE_C2[1] * s[5 % N], s6 = SHE_C2[2] * s[6 % N], s7 = SHE_C2[3] * s[7 % N],
                        s8 = SHE_C2[4] * s[8 % N];
            gx += s4 * y - 2.0f * s6 * x + s7 * z + 2.0f * s8 * x;
            gy += s4 * x + s5 * z - 2.0f * s6 * y - 2.0f * s8 * y;
            gz += s5 * y + 4.0f * s6 * z + s7 * x;
        }
        if (DEG > 2) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            const float s9 = SHE_C3[0] * s[9 % N], s10 = SHE_C3[1] * s[10 % N], s11 = SHE_C3[2] * s[11 % N], s12 = SHE_C3[3] * s[12 % N],
                        s13 = SHE_C3[4] * s[13 % N], s14 = SHE_C3[5] * s[14 % N], s15 = SHE_C3[6] * s[15 % N];
            gx += s9 * 6.0f * xy + s10 * yz - s11 * 2.0f * xy - s12 * 6.0f * xz + s13 * (4.0f * zz - 3.0f * xx - yy) + s14 * 2.0f * xz +
                  s15 * (3.0f * xx - 3.0f * yy);
            gy += s9 * (3.0f * xx - 3.0f * yy) + s10 * xz + s11 * (4.0f * zz - xx - 3.0f * yy) - s12 * 6.0f * yz - s13 * 2.0f * xy -
                  s14 * 2.0f * yz - s15 * 6.0f * xy;
            gz += s10 * xy + s11 * 8.0f * yz + s12 * (6.0f * zz - 3.0f * xx - 3.0f * yy) + s13 * 8.0f * xz + s14 * (xx - yy);
        }
        grad_dirs[i3] = gx;
        grad_dirs[i3 + 1] = gy;
        grad_dirs[i3 + 2] = gz;
    }
}

}  // namespace wg

extern "C" int wg_eval_sh_forward(int P, int deg, int K, const float* sh, const float* dirs, float* out, void* stream) {
    if (P < 0 || deg < 0 || deg > 3 || K < (deg + 1) * (deg + 1)) return WG_ERR_INVALID_ARGUMENT;
    if (P == 0) return WG_OK;
    if (!sh || !dirs || !out) return WG_ERR_INVALID_ARGUMENT;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const dim3 grid((P + 255) / 256), block(256);
    switch (deg) {
        case 0: wg::eval_sh_forward_kernel<0><<<grid, block, 0, s>>>(P, K, sh, dirs, out); break;
        case 1: wg::eval_sh_forward_kernel<1><<<grid, block, 0, s>>>(P, K, sh, dirs, out); break;
        case 2: wg::eval_sh_forward_kernel<2><<<grid, block, 0, s>>>(P, K, sh, dirs, out); break;
        default: wg::eval_sh_forward_kernel<3><<<grid, block, 0, s>>>(P, K, sh, dirs, out); break;
    }
    return hipGetLastError() == hipSuccess ? WG_OK : WG_ERR_HIP;
}

extern "C" int wg_eval_sh_backward(int P, int deg, int K, const float* sh, const float* dirs, const float* grad_out, float* grad_sh,
                                   float* grad_dirs, void* stream) {
    if (P < 0 || deg < 0 || deg > 3 || K < (deg + 1) * (deg + 1)) return WG_ERR_INVALID_ARGUMENT;
    if (P == 0) return WG_OK;
    if (!sh || !dirs || !grad_out || !grad_sh) return WG_ERR_INVALID_ARGUMENT;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const dim3 grid((P + 255) / 256), block(256);
    switch (deg) {
        case 0: wg::eval_sh_backward_kernel<0><<<grid, block, 0, s>>>(P, K, sh, dirs, grad_out, grad_sh, grad_dirs); break;
        case 1: wg::eval_sh_backward_kernel<1><<<grid, block, 0, s>>>(P, K, sh, dirs, grad_out, grad_sh, grad_dirs); break;
        case 2: wg::eval_sh_backward_kernel<2><<<grid, block, 0, s>>>(P, K, sh, dirs, grad_out, grad_sh, grad_dirs); break;
        default: wg::eval_sh_backward_kernel<3><<<grid, block, 0, s>>>(P, K, sh, dirs, grad_out, grad_sh, grad_dirs); break;
    }
    return hipGetLastError() == hipSuccess ? WG_OK : WG_ERR_HIP;
}
