// K9: per-tile back-to-front gradient walk for gfx950.  Replaces renderCUDA (backward.cu:435-606).
//
// Same wave-per-tile mapping as the forward kernel (render_fwd.hip): 64 lanes x 4 strips = 256 pixels.
// What changes relative to the reference's design:
//   * the walk starts at the tile's last contributing instance (tile_last, produced by the forward
//     kernel) instead of the end of the tile's list, so instances no pixel ever reached are never read;
//   * the reference issues 10 float atomicAdds to global memory per contributing (pixel, Gaussian) pair
//     (backward.cu:568-603).  Here each lane first sums its 4 pixels' contributions in registers, the
//     wave then reduces the 10 partial sums across its 64 lanes with DPP row operations (no LDS, no
//     barrier), and ONE lane issues the 10 hardware float atomics per (tile, Gaussian) instance:
//     256x fewer atomics than the reference for a fully covered tile;
//   * the "last_alpha / last_color" lazy recurrence (backward.cu:560-561,572) is applied eagerly
//     (accum_rec' = alpha*c + (1-alpha)*accum_rec right after use), the same values with 16 fewer
//     live registers per lane.
// The gradient arithmetic is the reference's hand-derived backward, not autograd of the forward:
// straight-through min(0.99,.), NDC-scaled dL_dmean2D (0.5*W, 0.5*H), abs-gradient in .z
// (backward.cu:593-595), conic gradient in .x/.y/.w of a float4 (backward.cu:598-600).
#include "wg_common.h"
#include "wg_alpha.h"

namespace wg {

constexpr int BATCH = 64;

// Sum over the 64 lanes of a wave; the total is valid in lane 63.
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
    // quad_perm [1,0,3,2], [2,3,0,1]; row_ror 4, 8; row_bcast 15 (rows 1,3), row_bcast 31 (rows 2,3)
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xF, 0xF, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xA, 0xF, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xC, 0xF, false));
    return v;
}

__global__ void __launch_bounds__(64) render_backward_kernel(
    int W, int H, int gx, int tiles, const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
    const float4* __restrict__ splats, const float2* __restrict__ subpixel_offset, const float* __restrict__ bg,
    const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib, const uint32_t* __restrict__ tile_last,
    const float* __restrict__ dL_dpix, float* __restrict__ dL_dmean2D, float* __restrict__ dL_dconic,
    float* __restrict__ dL_dopacity, float* __restrict__ dL_dcolor) {
    __shared__ float4 lds[BATCH * 3];
    __shared__ uint32_t lds_id[BATCH];

    const int tile = xcd_tile(blockIdx.x, tiles);
    const int hi0 = (int)tile_last[tile];
    if (hi0 == 0) return;
    const int lane = threadIdx.x;
    const int tx = tile % gx, ty = tile / gx;
    const int px = tx * TILE_X + (lane & 15);
    const int py0 = ty * TILE_Y + (lane >> 4);
    const size_t plane = (size_t)W * H;
    const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
    const float ddelx_dx = 0.5f * W, ddely_dy = 0.5f * H;

    float pfx[4], pfy[4], T[4], tfb[4], dLr[4], dLg[4], dLb[4], recr[4], recg[4], recb[4];
    int last[4];
#pragma unroll
    for (int s = 0; s < 4; s++) {
        const int py = py0 + 4 * s;
        const bool inside = px < W && py < H;
        float2 off = make_float2(0.f, 0.f);
        T[s] = 0.f; last[s] = 0; dLr[s] = dLg[s] = dLb[s] = 0.f;
        if (inside) {
            const size_t pix = (size_t)W * py + px;
            off = subpixel_offset[pix];
            T[s] = final_T[pix];
            last[s] = (int)n_contrib[pix];
            dLr[s] = dL_dpix[pix];
            dLg[s] = dL_dpix[plane + pix];
            dLb[s] = dL_dpix[2 * plane + pix];
        }
        pfx[s] = (float)px + off.x;
        pfy[s] = (float)py + off.y;
        tfb[s] = -T[s] * (bg0 * dLr[s] + bg1 * dLg[s] + bg2 * dLb[s]);  // -T_final * <bg, dL_dpixel>
        recr[s] = recg[s] = recb[s] = 0.f;
    }

    const uint2 range = ranges[tile];

    for (int hi = hi0; hi > 0; hi -= BATCH) {
        // lane l stages the instance at list position hi-1-l (back to front, backward.cu:517)
        const int posl = hi - 1 - lane;
        __syncthreads();
        if (posl >= 0) {
            const uint32_t id = point_list[range.x + posl];
            lds_id[lane] = id;
            lds[3 * lane] = splats[3 * (size_t)id];
            lds[3 * lane + 1] = splats[3 * (size_t)id + 1];
            lds[3 * lane + 2] = splats[3 * (size_t)id + 2];
        }
        __syncthreads();
        const int cnt = min(BATCH, hi);
        for (int j = 0; j < cnt; j++) {
            const int pos = hi - 1 - j;  // "contributor" after the decrement at backward.cu:531
            const float4 r0 = lds[3 * j];
            const float4 r1 = lds[3 * j + 1];
            const SplatCoef sc = make_coef(r0, r1);
            const float o = sc.o;
            float G[4], alpha[4], dxs[4], dys[4];
            uint32_t hit = 0;
#pragma unroll
            for (int s = 0; s < 4; s++) {
                const bool pass = eval_alpha(sc, pfx[s], pfy[s], dxs[s], dys[s], G[s], alpha[s]);
                if (pos < last[s] && pass) hit |= 1u << s;
            }
            if (__ballot(hit != 0) == 0ull) continue;

            const float cbch = lds[3 * j + 2].x;
            const float colr = r1.z, colg = r1.w;
            float acr = 0.f, acg = 0.f, acb = 0.f, amx = 0.f, amy = 0.f, aab = 0.f, axx = 0.f, axy = 0.f, ayy = 0.f, aop = 0.f;
#pragma unroll
            for (int s = 0; s < 4; s++) {
                if (__ballot((hit >> s) & 1u) == 0ull) continue;
                if ((hit >> s) & 1u) {
                    const float a = alpha[s];
                    const float inv = __builtin_amdgcn_rcpf(1.0f - a);
                    const float Tn = T[s] * inv;  // T / (1 - alpha), backward.cu:548
                    T[s] = Tn;
                    const float w = a * Tn;  // dchannel_dcolor
                    acr += w * dLr[s];
                    acg += w * dLg[s];
                    acb += w * dLb[s];
                    const float dr = colr - recr[s], dg = colg - recg[s], db = cbch - recb[s];
                    float dLda = dr * dLr[s] + dg * dLg[s] + db * dLb[s];
                    recr[s] += a * dr;  // accum_rec for the next (nearer) contributor
                    recg[s] += a * dg;
                    recb[s] += a * db;
                    dLda = dLda * Tn + tfb[s] * inv;
                    const float dLdG = o * dLda;
                    const float gdx = G[s] * dxs[s], gdy = G[s] * dys[s];
                    const float dGdx = -gdx * r0.z - gdy * r0.w;
                    const float dGdy = -gdy * r1.x - gdx * r0.w;
                    const float mx = dLdG * dGdx * ddelx_dx, my = dLdG * dGdy * ddely_dy;
                    amx += mx;
                    amy += my;
                    aab += fabsf(mx) + fabsf(my);
                    axx += -0.5f * gdx * dxs[s] * dLdG;
                    axy += -0.5f * gdx * dys[s] * dLdG;
                    ayy += -0.5f * gdy * dys[s] * dLdG;
                    aop += G[s] * dLda;
                }
            }
            acr = wave_sum_to_lane63(acr); acg = wave_sum_to_lane63(acg); acb = wave_sum_to_lane63(acb);
            amx = wave_sum_to_lane63(amx); amy = wave_sum_to_lane63(amy); aab = wave_sum_to_lane63(aab);
            axx = wave_sum_to_lane63(axx); axy = wave_sum_to_lane63(axy); ayy = wave_sum_to_lane63(ayy);
            aop = wave_sum_to_lane63(aop);
            if (lane == 63) {
                const size_t id = lds_id[j];
                unsafeAtomicAdd(dL_dcolor + 3 * id + 0, acr);
                unsafeAtomicAdd(dL_dcolor + 3 * id + 1, acg);
                unsafeAtomicAdd(dL_dcolor + 3 * id + 2, acb);
                unsafeAtomicAdd(dL_dmean2D + 3 * id + 0, amx);
                unsafeAtomicAdd(dL_dmean2D + 3 * id + 1, amy);
                unsafeAtomicAdd(dL_dmean2D + 3 * id + 2, aab);
                unsafeAtomicAdd(dL_dconic + 4 * id + 0, axx);
                unsafeAtomicAdd(dL_dconic + 4 * id + 1, axy);
                unsafeAtomicAdd(dL_dconic + 4 * id + 3, ayy);
                unsafeAtomicAdd(dL_dopacity + id, aop);
            }
        }
    }
}

hipError_t launch_render_backward(int W, int H, int gx, int gy, const ImageState& img, const BinningState& b,
                                  const GeometryState& g, const float* subpixel_offset, const float* background,
                                  const float* dL_dpix, float* dL_dmean2D, float* dL_dconic, float* dL_dopacity,
                                  float* dL_dcolor, hipStream_t stream) {
    const int tiles = gx * gy;
    if (tiles <= 0) return hipSuccess;
    hipLaunchKernelGGL(render_backward_kernel, dim3(tiles), dim3(64), 0, stream, W, H, gx, tiles, img.ranges, b.point_list, g.splats,
                       reinterpret_cast<const float2*>(subpixel_offset), background, img.final_T, img.n_contrib, img.tile_last,
                       dL_dpix, dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor);
    return hipGetLastError();
}

}  // namespace wg
