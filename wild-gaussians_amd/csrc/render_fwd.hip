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
//   * while staging, lane l also computes instance l's strip-reachability mask (wg_alpha.h: the 1/255
//     iso-ellipse's bounding box against the four strips' sample boxes).  The per-instance loop reads that
//     masks as four wave-uniform 64-bit ballots: unreachable strips and unreachable instances cost nothing.
//     The mask is conservative, so the blended result is unchanged.
//   * saturated strips (every pixel hit the T < 1e-4 stop) are dropped from the uniform mask; the walk ends
//     when no strip is left.
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
    int W, int H, int gx, int tiles, const uint32_t* __restrict__ order, const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
    const float4* __restrict__ splats, const float2* __restrict__ subpixel_offset, const float* __restrict__ bg,
    float* __restrict__ final_T, uint32_t* __restrict__ n_contrib, uint32_t* __restrict__ tile_last,
    float* __restrict__ out_color) {
    __shared__ float4 lds[BATCH * 3];

    const int tile = order ? (int)order[xcd_tile(blockIdx.x, tiles)] : xcd_tile(blockIdx.x, tiles);
    const int lane = threadIdx.x;
    const int tx = tile % gx, ty = tile / gx;
    const int px = tx * TILE_X + (lane & 15);
    const int py0 = ty * TILE_Y + (lane >> 4);

    float pfx[4], pfy[4], T[4], Cr[4], Cg[4], Cb[4];
    uint32_t last[4];
    uint32_t alive = 0;  // bit s set <=> pixel of strip s still accumulating
    StripBounds sb;
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
        const float inf = __builtin_huge_valf();
        sb.x0[s] = wave_min_uniform(inside ? pfx[s] : inf);
        sb.x1[s] = wave_max_uniform(inside ? pfx[s] : -inf);
        sb.y0[s] = wave_min_uniform(inside ? pfy[s] : inf);
        sb.y1[s] = wave_max_uniform(inside ? pfy[s] : -inf);
    }
    uint32_t strips_alive = 0;  // wave-uniform: strips with at least one unsaturated pixel
#pragma unroll
    for (int s = 0; s < 4; s++)
        if (__ballot((alive >> s) & 1u) != 0ull) strips_alive |= 1u << s;

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

    for (int base = 0; base < n && strips_alive != 0; base += BATCH) {
        const int cnt = min(BATCH, n - base);
        const uint32_t mymask = lane < cnt ? strip_mask(a0, a1, sb) : 0u;
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
        // The batch's strip masks as four wave-uniform 64-bit words (bit j of word s: instance j can reach strip s):
        // the walk below never touches an instance no live strip can see, and knows which strips to evaluate before
        // the instance's record has even been read.
        uint64_t reach[4];
#pragma unroll
        for (int s = 0; s < 4; s++) reach[s] = ((strips_alive >> s) & 1u) ? __ballot((mymask >> s) & 1u) : 0ull;
        uint64_t todo = reach[0] | reach[1] | reach[2] | reach[3];
        while (todo != 0ull) {
            const int j = __builtin_ctzll(todo);
            todo &= todo - 1;
            if (((reach[0] | reach[1] | reach[2] | reach[3]) >> j & 1ull) == 0ull) continue;  // its strips died meanwhile
            const float4 r0 = lds[3 * j];      // mx, my, conic.x, conic.y
            const float4 r1 = lds[3 * j + 1];  // conic.z, opacity, -, red
            const SplatCoef sc = make_coef(r0, r1);
            const uint32_t pos = (uint32_t)(base + j + 1);
            const uint32_t alive_before = alive;
#pragma unroll
            for (int s = 0; s < 4; s++) {
                if (((reach[s] >> j) & 1ull) == 0ull) continue;  // wave-uniform
                PairEval e;
                const bool pass = eval_alpha(sc, pfx[s], pfy[s], e);
                const float alpha = e.alpha;
                if (((alive >> s) & 1u) && pass) {
                    const float test_T = T[s] * (1.0f - alpha);
                    if (test_T < 0.0001f) {
                        alive &= ~(1u << s);  // done (forward.cu:368-372): this instance is not blended
                    } else {
                        const float2 gb = *reinterpret_cast<const float2*>(&lds[3 * j + 2]);
                        const float w = alpha * T[s];
                        Cr[s] += r1.w * w;
                        Cg[s] += gb.x * w;
                        Cb[s] += gb.y * w;
                        T[s] = test_T;
                        last[s] = pos;
                    }
                }
            }
            if (__ballot(alive != alive_before) != 0ull) {  // some pixel saturated: refresh the strip liveness
                strips_alive = 0;
#pragma unroll
                for (int s = 0; s < 4; s++) {
                    if (__ballot((alive >> s) & 1u) != 0ull) strips_alive |= 1u << s;
                    else reach[s] = 0ull;
                }
                if (strips_alive == 0) break;
            }
        }
    }

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
    hipLaunchKernelGGL(render_forward_kernel, dim3(tiles), dim3(64), 0, stream, W, H, gx, tiles, (const uint32_t*)nullptr, img.ranges, b.point_list, g.splats,
                       reinterpret_cast<const float2*>(subpixel_offset), background, img.final_T, img.n_contrib, img.tile_last,
                       out_color);
    return hipGetLastError();
}

}  // namespace wg
