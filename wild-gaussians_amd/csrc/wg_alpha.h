// Shared per-(pixel, splat) falloff evaluation, used by BOTH render kernels so that the forward and the
// backward pass take identical skip decisions (forward.cu:353-366 == backward.cu:536-546).
#pragma once
#include <hip/hip_runtime.h>

namespace wg {

constexpr float WG_LOG2E = 1.4426950408889634f;

struct SplatCoef {
    float mx, my;      // pixel-space mean
    float ca, cb, cc;  // log2(e) * (-0.5*conic.x, -conic.y, -0.5*conic.z)
    float o;           // opacity * mip-filter coef
};

__device__ __forceinline__ SplatCoef make_coef(const float4 r0, const float4 r1) {
    SplatCoef c;
    c.mx = r0.x;
    c.my = r0.y;
    c.ca = -0.5f * WG_LOG2E * r0.z;
    c.cb = -WG_LOG2E * r0.w;
    c.cc = -0.5f * WG_LOG2E * r1.x;
    c.o = r1.y;
    return c;
}

// Returns true when the pair passes the reference's two skips: power <= 0 and alpha >= 1/255.
// G = exp(power) (unclamped), alpha = min(0.99, o*G).
__device__ __forceinline__ bool eval_alpha(const SplatCoef& c, float pfx, float pfy, float& dx, float& dy, float& G, float& alpha) {
    dx = c.mx - pfx;
    dy = c.my - pfy;
    const float p2 = dx * (c.ca * dx + c.cb * dy) + c.cc * dy * dy;  // log2(e) * power
    G = __builtin_amdgcn_exp2f(p2);
    alpha = fminf(0.99f, c.o * G);
    return p2 <= 0.0f && alpha >= (1.0f / 255.0f);
}

// XCD-aware block -> tile map: blocks are dealt round-robin to the 8 XCDs (block b runs on XCD b % 8), so
// XCD x receives the contiguous band of tiles [x*q + min(x,rem), ...): its private L2 only ever sees the
// splat records of that band.  Bijective for any tile count.
__device__ __forceinline__ int xcd_tile(int b, int tiles) {
    const int q = tiles >> 3, rem = tiles & 7;
    const int xcd = b & 7, i = b >> 3;
    return xcd * q + min(xcd, rem) + i;
}

}  // namespace wg
