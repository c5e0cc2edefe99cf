// K8: per-tile front-to-back alpha compositing for gfx950.  Replaces renderCUDA (forward.cu:273-395).
//
// Mapping (wave64-first, not a 16x16-thread CUDA block):
//   * one 64-lane wave per 16x16 tile; lane l owns column x = l & 15 and rows (l >> 4) + 4*s for the
//     four 16x4 strips s = 0..3, i.e. 4 pixels per lane.  No workgroup barrier is ever needed (the
//     workgroup IS the wave), the per-instance LDS broadcast read is amortised over 256 pixel
//     evaluations, and each lane carries 4 independent dependency chains (ILP hides v_exp latency).
//   * the tile's sorted instance list is consumed in batches of 64: lane l gathers instance l's 48-byte
//     splat record (position, conic, opacity, colour -- one gather instead of the reference's four
//     arrays + a colour fetch from global per contributing pair, forward.cu:376) into registers while the
//     previous batch is being composited (software prefetch), then parks it in LDS.
//   * strips whose 64 lanes all miss an instance skip the blend with one wave-uniform branch; the walk
//     stops as soon as every pixel of the tile is saturated (checked per instance, not per 256-batch).
//   * block -> tile mapping is XCD-aware: blocks are dealt round-robin to the 8 XCDs, so XCD x is given a
//     contiguous band of tiles and its private 4 MiB L2 only has to hold that band's splat records.
//
// Arithmetic: alpha = min(0.99, o * exp(power)) is evaluated as exp2 of a pre-scaled quadratic form with
// v_exp_f32 and fused multiply-adds; thresholds (power > 0, alpha < 1/255, T*(1-alpha) < 1e-4) are the
// reference's (forward.cu:357-372).  Results agree with the literal float32 oracle to ~1e-6 except where a
// threshold decision sits within rounding distance (see tests/test_parity_gpu.py: fragile pixels).
#include "wg_common.h"
#include "wg_alpha.h"

namespace wg {

constexpr int BATCH = 64;

__global__ void __launch_bounds__(64) render_forward_kernel(
    int W, int H, int gx, int tiles, const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
    const float4* __restrict__ splats, const float2* __restrict__ subpixel_offset, const float* __restrict__ bg,
    float* __restrict__ final_T, uint32_t* __restrict__ n_contrib, uint32_t* __restrict__ tile_last,
    float* __restrict__ out_color) {
    __shared__ float4 lds[BATCH * 3];

    const int tile = xcd_tile(blockIdx.x, tiles);
    const int lane = threadIdx.x;
    const int tx = tile % gx, ty = tile / gx;
    const int px = tx * TILE_X + (lane & 15);
    const int py0 = ty * TILE_Y + (lane >> 4);

    float pfx[4], pfy[4], T[4], Cr[4], Cg[4], Cb[4];
    uint32_t last[4];
    uint32_t alive = 0;  // bit s set <=> pixel of strip s still accumulating
#pragma unroll
    for (int s = 0; s < 4; s++) {
        const int py = py0 + 4 * s;
        const bool inside = px < W && py < H;
        float2 off = make_float2(0.f, 0.f);
        if (inside) {
            off = subpixel_offset[(size_t)W * py + px];
            alive |= 1u << s;
        }
        pfx[s] = (float)px + off.x;
        pfy[s] = (float)py + off.y;
        T[s] = 1.0f;
        Cr[s] = Cg[s] = Cb[s] = 0.f;
        last[s] = 0;
    }

    const uint2 range = ranges[tile];
    const int n = (int)(range.y - range.x);

    float4 a0, a1, a2;
    a0 = a1 = a2 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (lane < n) {
        const uint32_t id = point_list[range.x + lane];
        a0 = splats[3 * (size_t)id];
        a1 = splats[3 * (size_t)id + 1];
        a2 = splats[3 * (size_t)id + 2];
    }

    for (int base = 0; base < n; base += BATCH) {
        __syncthreads();
        lds[3 * lane] = a0;
        lds[3 * lane + 1] = a1;
        lds[3 * lane + 2] = a2;
        __syncthreads();
        if (base + BATCH + lane < n) {  // prefetch the next batch under this batch's math
            const uint32_t id = point_list[range.x + base + BATCH + lane];
            a0 = splats[3 * (size_t)id];
            a1 = splats[3 * (size_t)id + 1];
            a2 = splats[3 * (size_t)id + 2];
        }
        const int cnt = min(BATCH, n - base);
        for (int j = 0; j < cnt; j++) {
            if (__ballot(alive != 0) == 0ull) goto finished;
            const float4 r0 = lds[3 * j];      // mx, my, conic.x, conic.y
            const float4 r1 = lds[3 * j + 1];  // conic.z, opacity, r, g
            const SplatCoef sc = make_coef(r0, r1);
            float alpha[4];
            uint32_t hit = 0;
#pragma unroll
            for (int s = 0; s < 4; s++) {
                float dx, dy, G;
                const bool pass = eval_alpha(sc, pfx[s], pfy[s], dx, dy, G, alpha[s]);
                if (((alive >> s) & 1u) && pass) hit |= 1u << s;
            }
            if (__ballot(hit != 0) == 0ull) continue;
            const float cbch = lds[3 * j + 2].x;
            const uint32_t pos = (uint32_t)(base + j + 1);
#pragma unroll
            for (int s = 0; s < 4; s++) {
                if (__ballot((hit >> s) & 1u) == 0ull) continue;
                if ((hit >> s) & 1u) {
                    const float test_T = T[s] * (1.0f - alpha[s]);
                    if (test_T < 0.0001f) {
                        alive &= ~(1u << s);  // done (forward.cu:368-372): this instance is not blended
                    } else {
                        const float w = alpha[s] * T[s];
                        Cr[s] += r1.z * w;
                        Cg[s] += r1.w * w;
                        Cb[s] += cbch * w;
                        T[s] = test_T;
                        last[s] = pos;
                    }
                }
            }
        }
    }
finished:
    uint32_t lmax = 0;
    const size_t plane = (size_t)W * H;
    const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
#pragma unroll
    for (int s = 0; s < 4; s++) {
        const int py = py0 + 4 * s;
        if (px < W && py < H) {
            const size_t pix = (size_t)W * py + px;
            final_T[pix] = T[s];
            n_contrib[pix] = last[s];
            out_color[pix] = Cr[s] + T[s] * bg0;
            out_color[plane + pix] = Cg[s] + T[s] * bg1;
            out_color[2 * plane + pix] = Cb[s] + T[s] * bg2;
            lmax = max(lmax, last[s]);
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) lmax = max(lmax, (uint32_t)__shfl_xor((int)lmax, m));
    if (lane == 0) tile_last[tile] = lmax;
}

hipError_t launch_render_forward(int W, int H, int gx, int gy, const ImageState& img, const BinningState& b,
                                 const GeometryState& g, const float* subpixel_offset, const float* background,
                                 float* out_color, hipStream_t stream) {
    const int tiles = gx * gy;
    if (tiles <= 0) return hipSuccess;
    hipLaunchKernelGGL(render_forward_kernel, dim3(tiles), dim3(64), 0, stream, W, H, gx, tiles, img.ranges, b.point_list, g.splats,
                       reinterpret_cast<const float2*>(subpixel_offset), background, img.final_T, img.n_contrib, img.tile_last,
                       out_color);
    return hipGetLastError();
}

}  // namespace wg
