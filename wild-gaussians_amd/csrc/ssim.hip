// Fused SSIM map, forward and backward, for gfx950 (include/wg_ssim.h).
// Reference semantics: wildgaussians/method.py:644-673 -- depthwise 11x11 Gaussian window (sigma 1.5, normalised 1-D
// taps, outer product), zero padding 5, C1 = 0.01^2, C2 = 0.03^2.
//
// HBM-bound stencil: the forward pass reads 8 B and writes 16 B per (channel, pixel), the backward pass reads 24 B and
// writes 4 B; everything in between lives in LDS.  One 256-thread workgroup owns a 32x16 output tile of one channel:
//   1. the 42x26 halo of both images goes to LDS (zero outside the frame = the reference's zero padding);
//   2. horizontal pass: the five running sums (x, y, xx, yy, xy) of the 11 taps for 26 rows x 32 columns -> LDS;
//   3. vertical pass: 11 taps over those rows -> mu1, mu2, E[xx], E[yy], E[xy] -> SSIM and its three partial derivatives.
// The backward pass is the same separable convolution applied to dL_dmap * (the three derivative maps).
#include <hip/hip_runtime.h>
#include <cmath>
#include <stddef.h>
#include "wg_ssim.h"
#include "wg_rasterizer.h"

namespace wg {

constexpr int SS_TW = 32, SS_TH = 16, SS_R = 5, SS_K = 11;
constexpr int SS_HW = SS_TW + 2 * SS_R, SS_HH = SS_TH + 2 * SS_R;  // 42 x 26 halo

struct SsimTaps {
    float w[SS_K];
};

// LOSS (include/wg_ssim.h: wg_l1_ssim_loss_*): instead of writing the SSIM map, the workgroup sums (1 - ssim) * mult and
// |img_l1 - img2| * mult over its tile (img2 = the ground truth) and leaves the two partial sums in partials[2 * block]; a second
// tiny kernel adds the partials in block order (deterministic) and forms the reference's loss (method.py:1948-1965).
template <bool LOSS>
__global__ void __launch_bounds__(256) ssim_forward_kernel(int H, int W, const float* __restrict__ img1, const float* __restrict__ img2,
                                                           float* __restrict__ ssim_map, float* __restrict__ dm_dmu1,
                                                           float* __restrict__ dm_dsigma1_sq, float* __restrict__ dm_dsigma12, SsimTaps taps,
                                                           const float* __restrict__ img_l1, const float* __restrict__ mult,
                                                           float* __restrict__ partials) {
    __shared__ float sx[SS_HH][SS_HW + 1], sy[SS_HH][SS_HW + 1];
    __shared__ float hs[5][SS_HH][SS_TW + 1];
    __shared__ float red[2][4];
    float acc_s = 0.f, acc_l = 0.f;
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * SS_TW, y0 = blockIdx.y * SS_TH;
    const size_t plane = (size_t)blockIdx.z * H * W;
    img1 += plane; img2 += plane;
    for (int i = tid; i < SS_HH * SS_HW; i += 256) {
        const int ly = i / SS_HW, lx = i % SS_HW;
        const int gy = y0 + ly - SS_R, gx = x0 + lx - SS_R;
        const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
        sx[ly][lx] = in ? img1[(size_t)gy * W + gx] : 0.f;
        sy[ly][lx] = in ? img2[(size_t)gy * W + gx] : 0.f;
    }
    __syncthreads();
    for (int i = tid; i < SS_HH * SS_TW; i += 256) {
        const int ly = i / SS_TW, lx = i % SS_TW;
        float a = 0.f, b = 0.f, aa = 0.f, bb = 0.f, ab = 0.f;
#pragma unroll
        for (int k = 0; k < SS_K; k++) {
            const float w = taps.w[k], u = sx[ly][lx + k], v = sy[ly][lx + k];
            a += w * u; b += w * v; aa += w * u * u; bb += w * v * v; ab += w * u * v;
        }
        hs[0][ly][lx] = a; hs[1][ly][lx] = b; hs[2][ly][lx] = aa; hs[3][ly][lx] = bb; hs[4][ly][lx] = ab;
    }
    __syncthreads();
    for (int i = tid; i < SS_TH * SS_TW; i += 256) {
        const int ly = i / SS_TW, lx = i % SS_TW;
        const int gy = y0 + ly, gx = x0 + lx;
        if (gy >= H || gx >= W) continue;
        float mu1 = 0.f, mu2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
        for (int k = 0; k < SS_K; k++) {
            const float w = taps.w[k];
            mu1 += w * hs[0][ly + k][lx]; mu2 += w * hs[1][ly + k][lx];
            e11 += w * hs[2][ly + k][lx]; e22 += w * hs[3][ly + k][lx]; e12 += w * hs[4][ly + k][lx];
        }
        const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
        const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
        const float s1 = e11 - mu1_sq, s2 = e22 - mu2_sq, s12 = e12 - mu12;
        const float A = 2.f * mu12 + C1, B = 2.f * s12 + C2, Cc = mu1_sq + mu2_sq + C1, D = s1 + s2 + C2;
        const size_t o = plane + (size_t)gy * W + gx;
        const float m = (A * B) / (Cc * D);
        if (LOSS) {
            const float wgt = mult ? mult[(size_t)gy * W + gx] : 1.0f;
            acc_s += (1.0f - m) * wgt;
            acc_l += fabsf(img_l1[o] - sy[ly + SS_R][lx + SS_R]) * wgt;
        } else {
            ssim_map[o] = m;
        }
        if (dm_dmu1) {
            // total derivative w.r.t. mu1, including sigma1_sq = E[xx] - mu1^2 and sigma12 = E[xy] - mu1*mu2
            const float iCD = 1.f / (Cc * D);
            dm_dmu1[o] = 2.f * mu2 * B * iCD - 2.f * mu2 * A * iCD - 2.f * mu1 * A * B / (Cc * Cc * D) + 2.f * mu1 * A * B / (Cc * D * D);
            dm_dsigma1_sq[o] = -A * B / (Cc * D * D);
            dm_dsigma12[o] = 2.f * A * iCD;
        }
    }
    if (LOSS) {  // workgroup sum in a fixed order: lanes (xor butterfly), then the four waves
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            acc_s += __shfl_xor(acc_s, d);
            acc_l += __shfl_xor(acc_l, d);
        }
        if ((tid & 63) == 0) { red[0][tid >> 6] = acc_s; red[1][tid >> 6] = acc_l; }
        __syncthreads();
        if (tid == 0) {
            const size_t b = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
            partials[2 * b] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
            partials[2 * b + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
        }
    }
}

// loss_out = { (1 - lambda) * l1_mean + lambda * dssim_mean, l1_mean, ssim_mean } from the per-workgroup partial sums
__global__ void __launch_bounds__(1024) l1_ssim_finish_kernel(const float* __restrict__ partials, int nblocks, float inv_count, float lambda,
                                                              float* __restrict__ loss_out) {
    __shared__ double red[2][16];
    double s = 0.0, l = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 1024) { s += partials[2 * i]; l += partials[2 * i + 1]; }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        s += __shfl_xor(s, d);
        l += __shfl_xor(l, d);
    }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s; red[1][threadIdx.x >> 6] = l; }
    __syncthreads();
    if (threadIdx.x == 0) {
        s = 0.0; l = 0.0;
        for (int w = 0; w < 16; w++) { s += red[0][w]; l += red[1][w]; }
        const float dssim = (float)(s * inv_count), l1 = (float)(l * inv_count);
        loss_out[0] = (1.0f - lambda) * l1 + lambda * dssim;
        loss_out[1] = l1;
        loss_out[2] = 1.0f - dssim;   // mean SSIM when mult == 1 (what the reference logs, method.py:1972)
    }
}

// dL_dimg1(q) = sum_p w(p - q) g(p) [ dm_dmu1(p) + 2 img1(q) dm_dsigma1_sq(p) + img2(q) dm_dsigma12(p) ]
// LOSS: dL_dmap is not a tensor but  -lambda * inv_count * (*dL_dloss) * mult[pixel]  (the loss is lambda * mean((1 - ssim) mult)),
// and the L1 branch's gradient  (1 - lambda) * inv_count * (*dL_dloss) * mult * sign(img_l1 - img2)  is written alongside.
template <bool LOSS>
__global__ void __launch_bounds__(256) ssim_backward_kernel(int H, int W, const float* __restrict__ img1, const float* __restrict__ img2,
                                                            const float* __restrict__ dL_dmap, const float* __restrict__ dm_dmu1,
                                                            const float* __restrict__ dm_dsigma1_sq, const float* __restrict__ dm_dsigma12,
                                                            float* __restrict__ dL_dimg1, SsimTaps taps, const float* __restrict__ img_l1,
                                                            const float* __restrict__ mult, const float* __restrict__ dL_dloss, float lambda,
                                                            float inv_count, float* __restrict__ dL_dimg_l1) {
    const float gl = LOSS ? dL_dloss[0] * inv_count : 0.f;
    __shared__ float t[3][SS_HH][SS_HW + 1];
    __shared__ float hs[3][SS_HH][SS_TW + 1];
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * SS_TW, y0 = blockIdx.y * SS_TH;
    const size_t plane = (size_t)blockIdx.z * H * W;
    for (int i = tid; i < SS_HH * SS_HW; i += 256) {
        const int ly = i / SS_HW, lx = i % SS_HW;
        const int gy = y0 + ly - SS_R, gx = x0 + lx - SS_R;
        float a = 0.f, b = 0.f, c = 0.f;
        if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
            const size_t o = plane + (size_t)gy * W + gx;
            const float g = LOSS ? -lambda * gl * (mult ? mult[(size_t)gy * W + gx] : 1.0f) : dL_dmap[o];
            a = g * dm_dmu1[o]; b = g * dm_dsigma1_sq[o]; c = g * dm_dsigma12[o];
        }
        t[0][ly][lx] = a; t[1][ly][lx] = b; t[2][ly][lx] = c;
    }
    __syncthreads();
    for (int i = tid; i < SS_HH * SS_TW; i += 256) {
        const int ly = i / SS_TW, lx = i % SS_TW;
        float a = 0.f, b = 0.f, c = 0.f;
#pragma unroll
        for (int k = 0; k < SS_K; k++) {
            const float w = taps.w[k];
            a += w * t[0][ly][lx + k]; b += w * t[1][ly][lx + k]; c += w * t[2][ly][lx + k];
        }
        hs[0][ly][lx] = a; hs[1][ly][lx] = b; hs[2][ly][lx] = c;
    }
    __syncthreads();
    for (int i = tid; i < SS_TH * SS_TW; i += 256) {
        const int ly = i / SS_TW, lx = i % SS_TW;
        const int gy = y0 + ly, gx = x0 + lx;
        if (gy >= H || gx >= W) continue;
        float a = 0.f, b = 0.f, c = 0.f;
#pragma unroll
        for (int k = 0; k < SS_K; k++) {
            const float w = taps.w[k];
            a += w * hs[0][ly + k][lx]; b += w * hs[1][ly + k][lx]; c += w * hs[2][ly + k][lx];
        }
        const size_t o = plane + (size_t)gy * W + gx;
        const float gt = img2[o];
        const float gs = a + 2.f * img1[o] * b + gt * c;
        if (LOSS) {
            const float d = img_l1[o] - gt;
            const float gl1 = (1.0f - lambda) * gl * (mult ? mult[(size_t)gy * W + gx] : 1.0f) * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
            if (dL_dimg_l1 == dL_dimg1) {  // one image feeds both terms: a single gradient
                dL_dimg1[o] = gs + gl1;
            } else {
                dL_dimg1[o] = gs;
                dL_dimg_l1[o] = gl1;
            }
        } else {
            dL_dimg1[o] = gs;
        }
    }
}

static SsimTaps make_taps() {  // method.py:648-649: exp(-(x - 5)^2 / (2 sigma^2)), normalised, in float32 like torch.Tensor
    SsimTaps t;
    float sum = 0.f;
    for (int k = 0; k < SS_K; k++) {
        t.w[k] = (float)std::exp(-(double)((k - SS_R) * (k - SS_R)) / (2.0 * 1.5 * 1.5));
        sum += t.w[k];
    }
    for (int k = 0; k < SS_K; k++) t.w[k] /= sum;
    return t;
}

}  // namespace wg

extern "C" {

int wg_ssim_forward(int C, int H, int W, const float* img1, const float* img2, float* ssim_map, float* dm_dmu1,
                    float* dm_dsigma1_sq, float* dm_dsigma12, void* stream) {
    if (C <= 0 || H <= 0 || W <= 0 || !img1 || !img2 || !ssim_map) return WG_ERR_INVALID_ARGUMENT;
    if ((dm_dmu1 != nullptr) != (dm_dsigma1_sq != nullptr) || (dm_dmu1 != nullptr) != (dm_dsigma12 != nullptr)) return WG_ERR_INVALID_ARGUMENT;
    const dim3 grid((W + wg::SS_TW - 1) / wg::SS_TW, (H + wg::SS_TH - 1) / wg::SS_TH, C);
    hipLaunchKernelGGL(wg::ssim_forward_kernel<false>, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream), H, W, img1, img2, ssim_map,
                       dm_dmu1, dm_dsigma1_sq, dm_dsigma12, wg::make_taps(), (const float*)nullptr, (const float*)nullptr, (float*)nullptr);
    return hipGetLastError() == hipSuccess ? WG_OK : WG_ERR_HIP;
}

int wg_ssim_backward(int C, int H, int W, const float* img1, const float* img2, const float* dL_dmap, const float* dm_dmu1,
                     const float* dm_dsigma1_sq, const float* dm_dsigma12, float* dL_dimg1, void* stream) {
    if (C <= 0 || H <= 0 || W <= 0 || !img1 || !img2 || !dL_dmap || !dm_dmu1 || !dm_dsigma1_sq || !dm_dsigma12 || !dL_dimg1)
        return WG_ERR_INVALID_ARGUMENT;
    const dim3 grid((W + wg::SS_TW - 1) / wg::SS_TW, (H + wg::SS_TH - 1) / wg::SS_TH, C);
    hipLaunchKernelGGL(wg::ssim_backward_kernel<false>, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream), H, W, img1, img2, dL_dmap,
                       dm_dmu1, dm_dsigma1_sq, dm_dsigma12, dL_dimg1, wg::make_taps(), (const float*)nullptr, (const float*)nullptr,
                       (const float*)nullptr, 0.f, 0.f, (float*)nullptr);
    return hipGetLastError() == hipSuccess ? WG_OK : WG_ERR_HIP;
}

size_t wg_l1_ssim_loss_scratch_floats(int C, int H, int W) {
    if (C <= 0 || H <= 0 || W <= 0) return 0;
    return 2 * (size_t)((W + wg::SS_TW - 1) / wg::SS_TW) * (size_t)((H + wg::SS_TH - 1) / wg::SS_TH) * (size_t)C;
}

int wg_l1_ssim_loss_forward(int C, int H, int W, const float* img_l1, const float* img_ssim, const float* gt, const float* mult,
                            float lambda, float* scratch, float* loss_out, float* dm_dmu1, float* dm_dsigma1_sq, float* dm_dsigma12,
                            void* stream) {
    if (C <= 0 || H <= 0 || W <= 0 || !img_l1 || !img_ssim || !gt || !scratch || !loss_out) return WG_ERR_INVALID_ARGUMENT;
    if ((dm_dmu1 != nullptr) != (dm_dsigma1_sq != nullptr) || (dm_dmu1 != nullptr) != (dm_dsigma12 != nullptr)) return WG_ERR_INVALID_ARGUMENT;
    const dim3 grid((W + wg::SS_TW - 1) / wg::SS_TW, (H + wg::SS_TH - 1) / wg::SS_TH, C);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(wg::ssim_forward_kernel<true>, grid, dim3(256), 0, st, H, W, img_ssim, gt, (float*)nullptr, dm_dmu1, dm_dsigma1_sq,
                       dm_dsigma12, wg::make_taps(), img_l1, mult, scratch);
    hipLaunchKernelGGL(wg::l1_ssim_finish_kernel, dim3(1), dim3(1024), 0, st, scratch, (int)(grid.x * grid.y * grid.z),
                       1.0f / ((float)C * (float)H * (float)W), lambda, loss_out);
    return hipGetLastError() == hipSuccess ? WG_OK : WG_ERR_HIP;
}

int wg_l1_ssim_loss_backward(int C, int H, int W, const float* img_l1, const float* img_ssim, const float* gt, const float* mult,
                             float lambda, const float* dL_dloss, const float* dm_dmu1, const float* dm_dsigma1_sq,
                             const float* dm_dsigma12, float* dL_dimg_l1, float* dL_dimg_ssim, void* stream) {
    if (C <= 0 || H <= 0 || W <= 0 || !img_l1 || !img_ssim || !gt || !dL_dloss || !dm_dmu1 || !dm_dsigma1_sq || !dm_dsigma12 ||
        !dL_dimg_l1 || !dL_dimg_ssim)
        return WG_ERR_INVALID_ARGUMENT;
    if ((img_l1 == img_ssim) != (dL_dimg_l1 == dL_dimg_ssim)) return WG_ERR_INVALID_ARGUMENT;  // one image <-> one gradient
    const dim3 grid((W + wg::SS_TW - 1) / wg::SS_TW, (H + wg::SS_TH - 1) / wg::SS_TH, C);
    hipLaunchKernelGGL(wg::ssim_backward_kernel<true>, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream), H, W, img_ssim, gt,
                       (const float*)nullptr, dm_dmu1, dm_dsigma1_sq, dm_dsigma12, dL_dimg_ssim, wg::make_taps(), img_l1, mult, dL_dloss, lambda,
                       1.0f / ((float)C * (float)H * (float)W), dL_dimg_l1);
    return hipGetLastError() == hipSuccess ? WG_OK : WG_ERR_HIP;
}

}  // extern "C"
