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

// d/d(x, y, z) of sum_k s_k basis_k(x, y, z): the polynomials differentiated with x, y, z as independent variables
template <int DEG>
__device__ __forceinline__ void sh_dir_grad(float x, float y, float z, const float* s, float& gx, float& gy, float& gz) {
    constexpr int N = (DEG + 1) * (DEG + 1);
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

// K == 16 (the reference's layout): a wave's 64 Gaussians are one contiguous 12 KB block of `sh`.  It is moved with coalesced
// 16-byte accesses (lane l takes float4 number 64 t + l of the block, t = 0..11) and transposed through LDS -- pitch 13 float4 per
// Gaussian, conflict-free for b128 -- instead of 12 loads at a 192-byte lane stride (the same scheme as the preprocess kernel's SH
// block).
constexpr int SHE_ROWS = 12, SHE_PITCH = 13;

__device__ __forceinline__ void she_block_to_lds(const float* __restrict__ sh, size_t first, int valid, int lane, float4* lds) {
    const float4* src = reinterpret_cast<const float4*>(sh) + first * SHE_ROWS;
#pragma unroll
    for (int t = 0; t < SHE_ROWS; t++) {
        const int idx = t * 64 + lane;
        if (idx < valid * SHE_ROWS) lds[(idx / SHE_ROWS) * SHE_PITCH + idx % SHE_ROWS] = src[idx];
    }
}
__device__ __forceinline__ void she_lds_to_block(float* __restrict__ dst_base, size_t first, int valid, int lane, const float4* lds) {
    float4* dst = reinterpret_cast<float4*>(dst_base) + first * SHE_ROWS;
#pragma unroll
    for (int t = 0; t < SHE_ROWS; t++) {
        const int idx = t * 64 + lane;
        if (idx < valid * SHE_ROWS) dst[idx] = lds[(idx / SHE_ROWS) * SHE_PITCH + idx % SHE_ROWS];
    }
}

template <int DEG>
__global__ void __launch_bounds__(256) eval_sh_forward_k16_kernel(int P, const float* __restrict__ sh, const float* __restrict__ dirs,
                                                                  float* __restrict__ out) {
    constexpr int N = (DEG + 1) * (DEG + 1);
    __shared__ float4 lds_all[4 * 64 * SHE_PITCH];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float4* lds = lds_all + wave * 64 * SHE_PITCH;
    const size_t first = (size_t)blockIdx.x * 256 + wave * 64;  // the wave's first Gaussian
    if (first >= (size_t)P) return;
    const int valid = (int)min((size_t)64, (size_t)P - first);
    she_block_to_lds(sh, first, valid, lane, lds);
    __builtin_amdgcn_wave_barrier();  // the wave is the only user of its LDS region; its LDS operations execute in issue order
    if (lane >= valid) return;
    const size_t i3 = 3 * (first + lane);
    float b[N];
    sh_basis<DEG>(dirs[i3], dirs[i3 + 1], dirs[i3 + 2], b);
#pragma unroll
    for (int c = 0; c < 3; c++) {
        float v[16];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const float4 t = lds[lane * SHE_PITCH + c * 4 + q];
            v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
        }
        float r = b[0] * v[0];
#pragma unroll
        for (int k = 1; k < N; k++) r += b[k] * v[k];
        out[i3 + c] = r;
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
__global__ void __launch_bounds__(256) eval_sh_backward_k16_kernel(int P, const float* __restrict__ sh, const float* __restrict__ dirs,
                                                                   const float* __restrict__ grad_out, float* __restrict__ grad_sh,
                                                                   float* __restrict__ grad_dirs) {
    constexpr int N = (DEG + 1) * (DEG + 1);
    __shared__ float4 lds_all[4 * 64 * SHE_PITCH];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float4* lds = lds_all + wave * 64 * SHE_PITCH;
    const size_t first = (size_t)blockIdx.x * 256 + wave * 64;
    if (first >= (size_t)P) return;
    const int valid = (int)min((size_t)64, (size_t)P - first);
    const bool want_dirs = grad_dirs != nullptr && DEG > 0;
    if (want_dirs) she_block_to_lds(sh, first, valid, lane, lds);
    __builtin_amdgcn_wave_barrier();
    const bool active = lane < valid;
    const size_t i3 = 3 * (first + (active ? lane : 0));
    const float x = dirs[i3], y = dirs[i3 + 1], z = dirs[i3 + 2];
    float b[N];
    sh_basis<DEG>(x, y, z, b);
    const float g[3] = {grad_out[i3], grad_out[i3 + 1], grad_out[i3 + 2]};
    float s[N];
#pragma unroll
    for (int k = 0; k < N; k++) s[k] = 0.f;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        if (want_dirs) {
#pragma unroll
            for (int q = 0; q < (N + 3) / 4; q++) {
                const float4 t = lds[lane * SHE_PITCH + c * 4 + q];
                if (4 * q < N) s[4 * q < N ? 4 * q : 0] += g[c] * t.x;
                if (4 * q + 1 < N) s[4 * q + 1 < N ? 4 * q + 1 : 0] += g[c] * t.y;
                if (4 * q + 2 < N) s[4 * q + 2 < N ? 4 * q + 2 : 0] += g[c] * t.z;
                if (4 * q + 3 < N) s[4 * q + 3 < N ? 4 * q + 3 : 0] += g[c] * t.w;
            }
        }
    }
    __builtin_amdgcn_wave_barrier();  // every lane has read its coefficients: the region is reused for the gradient rows
#pragma unroll
    for (int c = 0; c < 3; c++) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            float4 t;
            t.x = 4 * q < N ? b[4 * q < N ? 4 * q : 0] * g[c] : 0.f;
            t.y = 4 * q + 1 < N ? b[4 * q + 1 < N ? 4 * q + 1 : 0] * g[c] : 0.f;
            t.z = 4 * q + 2 < N ? b[4 * q + 2 < N ? 4 * q + 2 : 0] * g[c] : 0.f;
            t.w = 4 * q + 3 < N ? b[4 * q + 3 < N ? 4 * q + 3 : 0] * g[c] : 0.f;
            lds[lane * SHE_PITCH + c * 4 + q] = t;
        }
    }
    __builtin_amdgcn_wave_barrier();
    she_lds_to_block(grad_sh, first, valid, lane, lds);
    if (grad_dirs && active) {
        float gx = 0.f, gy = 0.f, gz = 0.f;
        sh_dir_grad<DEG>(x, y, z, s, gx, gy, gz);
        grad_dirs[i3] = gx;
        grad_dirs[i3 + 1] = gy;
        grad_dirs[i3 + 2] = gz;
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
        sh_dir_grad<DEG>(x, y, z, s, gx, gy, gz);
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
    if (K == 16 && (reinterpret_cast<uintptr_t>(sh) & 15u) == 0) {
        switch (deg) {
            case 0: wg::eval_sh_forward_k16_kernel<0><<<grid, block, 0, s>>>(P, sh, dirs, out); break;
            case 1: wg::eval_sh_forward_k16_kernel<1><<<grid, block, 0, s>>>(P, sh, dirs, out); break;
            case 2: wg::eval_sh_forward_k16_kernel<2><<<grid, block, 0, s>>>(P, sh, dirs, out); break;
            default: wg::eval_sh_forward_k16_kernel<3><<<grid, block, 0, s>>>(P, sh, dirs, out); break;
        }
        return hipGetLastError() == hipSuccess ? WG_OK : WG_ERR_HIP;
    }
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
    if (K == 16 && ((reinterpret_cast<uintptr_t>(sh) | reinterpret_cast<uintptr_t>(grad_sh)) & 15u) == 0) {
        switch (deg) {
            case 0: wg::eval_sh_backward_k16_kernel<0><<<grid, block, 0, s>>>(P, sh, dirs, grad_out, grad_sh, grad_dirs); break;
            case 1: wg::eval_sh_backward_k16_kernel<1><<<grid, block, 0, s>>>(P, sh, dirs, grad_out, grad_sh, grad_dirs); break;
            case 2: wg::eval_sh_backward_k16_kernel<2><<<grid, block, 0, s>>>(P, sh, dirs, grad_out, grad_sh, grad_dirs); break;
            default: wg::eval_sh_backward_k16_kernel<3><<<grid, block, 0, s>>>(P, sh, dirs, grad_out, grad_sh, grad_dirs); break;
        }
        return hipGetLastError() == hipSuccess ? WG_OK : WG_ERR_HIP;
    }
    switch (deg) {
        case 0: wg::eval_sh_backward_kernel<0><<<grid, block, 0, s>>>(P, K, sh, dirs, grad_out, grad_sh, grad_dirs); break;
        case 1: wg::eval_sh_backward_kernel<1><<<grid, block, 0, s>>>(P, K, sh, dirs, grad_out, grad_sh, grad_dirs); break;
        case 2: wg::eval_sh_backward_kernel<2><<<grid, block, 0, s>>>(P, K, sh, dirs, grad_out, grad_sh, grad_dirs); break;
        default: wg::eval_sh_backward_kernel<3><<<grid, block, 0, s>>>(P, K, sh, dirs, grad_out, grad_sh, grad_dirs); break;
    }
    return hipGetLastError() == hipSuccess ? WG_OK : WG_ERR_HIP;
}
