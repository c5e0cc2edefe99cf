// K8: per-tile front-to-back alpha compositing for gfx950.  Replaces renderCUDA (forward.cu:273-395).
//
// Mapping (wave64-first, not a 16x16-thread CUDA block):
//   * one 64-lane wave per 16x16 tile; lane l owns one pixel in each of the tile's four strips s = 0..3 (the 8x8 quadrants:
//     x = 8 (s & 1) + (l & 7), y = 8 (s >> 1) + (l >> 3); wg_alpha.h: strip_x / strip_y), i.e. 4 pixels per lane.  No workgroup barrier is ever needed (the
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
#include "wg_sort.h"

namespace wg {

constexpr int BATCH = 64;

// WG_COUNT_PAIRS (a VARIANT build only: scripts/count_pairs.py; never the product library): what the walk does, counted -- the inputs of
// SURVEY 8(d)'s secondary (compute) ceiling.  [0] instances visited, [1] strip evaluations (one = 64 (pixel, entry) pairs), [2] of those
// pairs, the ones whose pixel is still accumulating, [3] pairs that pass both skips, [4] pixels stopped by the T < 1e-4 test (a pair that
// passes is either blended or stops its pixel: blended = [3] - [4]).  (The counters live in lane 0's registers: they are only touched
// where every lane is active.)
#ifndef WG_COUNT_PAIRS
#define WG_COUNT_PAIRS 0
#endif
#if WG_COUNT_PAIRS
__device__ unsigned long long g_fwd_counters[8];
#define WG_CNT(i, v) wgc[i] += (unsigned long long)(v)
#else
#define WG_CNT(i, v)
#endif

// ---- the per-tile walk, as device functions shared by render_forward_kernel and the lazy-sort fix-up kernel (binning.hip) ----
// One wave owns a tile.  The walk is resumable: it composites list positions [pos_begin, pos_end) and can be continued
// later from the state parked in the output buffers (fwd_store with complete == false / fwd_init with resume == true).

// DUAL (two colour sets composited in ONE walk: wg_second_image in include/wg_rasterizer.h; WildGaussians renders raw and toned colours over
// identical geometry, method.py:1573-1611): the splat record's three spare floats (r1.z, r2.z, r2.w) carry the second set, the tile
// state three more sums per pixel, out_color2 receives the second image.  Every decision (alpha, T, n_contrib) is shared.
template <bool DUAL = false>
__device__ void fwd_init(FwdTile& st, int W, int H, int gx, int tile, int lane, const float2* __restrict__ subpixel_offset, bool resume,
                         const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib, const float* __restrict__ out_color,
                         const float* __restrict__ out_color2 = nullptr) {
    const int tx = tile % gx, ty = tile / gx;
    st.x0 = tx * TILE_X;
    st.y0 = ty * TILE_Y;
    st.alive = 0;
    const size_t plane = (size_t)W * H;
#pragma unroll
    for (int s = 0; s < 4; s++) {
        const int px = st.x0 + strip_x(lane, s), py = st.y0 + strip_y(lane, s);
        const bool inside = px < W && py < H;
        float2 off = make_float2(0.f, 0.f);
        st.T[s] = 1.0f;
        st.Cr[s] = st.Cg[s] = st.Cb[s] = 0.f;
        if (DUAL) st.C2r[s] = st.C2g[s] = st.C2b[s] = 0.f;
        st.last[s] = 0;
        if (inside) {
            const size_t pix = (size_t)W * py + px;
            if (subpixel_offset) off = subpixel_offset[pix];
            st.alive |= 1u << s;
            if (resume) {  // parked state: T < 0 marks a pixel that already hit the T < 1e-4 stop
                const float t = final_T[pix];
                st.T[s] = fabsf(t);
                if (t < 0.f) st.alive &= ~(1u << s);
                st.last[s] = n_contrib[pix];
                st.Cr[s] = out_color[pix];
                st.Cg[s] = out_color[plane + pix];
                st.Cb[s] = out_color[2 * plane + pix];
                if (DUAL) {
                    st.C2r[s] = out_color2[pix];
                    st.C2g[s] = out_color2[plane + pix];
                    st.C2b[s] = out_color2[2 * plane + pix];
                }
            }
        }
        st.pfx[s] = (float)px + off.x;
        st.pfy[s] = (float)py + off.y;
        const float inf = __builtin_huge_valf();
        st.sb.x0[s] = wave_min_uniform(inside ? st.pfx[s] : inf);
        st.sb.x1[s] = wave_max_uniform(inside ? st.pfx[s] : -inf);
        st.sb.y0[s] = wave_min_uniform(inside ? st.pfy[s] : inf);
        st.sb.y1[s] = wave_max_uniform(inside ? st.pfy[s] : -inf);
    }
    st.strips_alive = 0;  // wave-uniform: strips with at least one unsaturated pixel
#pragma unroll
    for (int s = 0; s < 4; s++)
        if (__ballot((st.alive >> s) & 1u) != 0ull) st.strips_alive |= 1u << s;
}

// lds: BATCH * 3 float4 of the wave's own LDS.  list = the tile's sorted instance list (point_list + range.x).
// EXACT (Options::exact_compositing): every value a skip / stop decision is taken on -- power, alpha, T -- is computed with the
// reference's own float32 operations (wg_alpha.h: eval_alpha_exact), so n_contrib, final_T and the set of blended instances are the
// reference's bit for bit; only the colour sums keep their fused multiply-adds (<= 1e-6 of a pixel).
template <bool WGB, bool EXACT, bool DUAL = false>
__device__ __forceinline__ void fwd_walk(FwdTile& st, float4* lds, int lane, const uint32_t* __restrict__ list, const float4* __restrict__ splats,
                         int pos_begin, int pos_end) {
    const int n = pos_end - pos_begin;
    list += pos_begin;
    float pfx[4], pfy[4], T[4], Cr[4], Cg[4], Cb[4], C2r[4], C2g[4], C2b[4];
    uint32_t last[4];
#pragma unroll
    for (int s = 0; s < 4; s++) {
        pfx[s] = st.pfx[s]; pfy[s] = st.pfy[s]; T[s] = st.T[s]; Cr[s] = st.Cr[s]; Cg[s] = st.Cg[s]; Cb[s] = st.Cb[s];
        last[s] = st.last[s];
        if (DUAL) { C2r[s] = st.C2r[s]; C2g[s] = st.C2g[s]; C2b[s] = st.C2b[s]; }
    }
    // Which pixels are still accumulating, as four WAVE-UNIFORM 64-bit masks (bit l of alive_m[s]: lane l's pixel of strip s): they live in
    // scalar registers, a strip evaluation applies one with a scalar AND on the execution mask (inverse_ballot) instead of two vector
    // instructions on a per-lane bit field, and the liveness of the strips is scalar arithmetic (round 5: -2 VALU of ~37 per strip evaluation).
    uint64_t alive_m[4];
#pragma unroll
    for (int s = 0; s < 4; s++) alive_m[s] = __builtin_amdgcn_ballot_w64(((st.alive >> s) & 1u) != 0u);
    uint32_t strips_alive = st.strips_alive;
    const StripBounds sb = st.sb;

    // Records travel list -> registers -> LDS one batch ahead of the walk, their ids two batches ahead, so that neither of the two
    // dependent global loads is waited for while it is in flight.  Indices are clamped to the list instead of predicated (lanes past
    // the end re-read its last entry: one more cache hit): with a conditional load the loop-carried registers are a merge of "loaded"
    // and "kept", and hipcc resolved that merge with copies placed right behind the loads -- i.e. an s_waitcnt that stalled the wave
    // on the "prefetch" it had just issued.
    if (n <= 0 || strips_alive == 0) return;  // nothing to do: st is unchanged
#if WG_COUNT_PAIRS
    unsigned long long wgc[6] = {0, 0, 0, 0, 0, 0};
#endif
    const int nl = n - 1;
    float4 a0, a1, a2;
    {
        const size_t r = 3 * (size_t)list[min(lane, nl)];
        a0 = splats[r];
        a1 = splats[r + 1];
        a2 = splats[r + 2];
    }
    uint32_t id_next = list[min(BATCH + lane, nl)];

    for (int base = 0; base < n && strips_alive != 0; base += BATCH) {
        const int cnt = min(BATCH, n - base);
        float4 s0 = a0, s1 = a1;
        const float4 s2 = a2;
        const uint32_t mymask = lane < cnt ? strip_mask(s0, s1, s2, sb) : 0u;
        // The wave is the staging area's only user and its LDS operations execute in issue order, so no barrier is needed for
        // correctness.  WGB (legal only when the workgroup IS the wave) keeps the s_barrier-free __syncthreads() of the one-wave
        // kernel anyway: hipcc schedules and allocates the compositing loop measurably better around it (0.325 vs 0.340 ms).
        if (WGB) __syncthreads(); else __builtin_amdgcn_wave_barrier();
        if (EXACT) halve_conic(s0, s1); else scale_conic(s0, s1);
        lds[3 * lane] = s0;
        lds[3 * lane + 1] = s1;
        lds[3 * lane + 2] = s2;
        if (WGB) __syncthreads(); else __builtin_amdgcn_wave_barrier();
        {
            const size_t r = 3 * (size_t)id_next;
            a0 = splats[r];
            a1 = splats[r + 1];
            a2 = splats[r + 2];
            id_next = list[min(base + 2 * BATCH + lane, nl)];
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
            const float4 r0 = lds[3 * j];      // mx, my, ca, cb        (EXACT: mx, my, -0.5 conic.x, conic.y)
            const float4 r1 = lds[3 * j + 1];  // cc, opacity, -, red   (EXACT: -0.5 conic.z, ...)
            const SplatCoef sc = coef_of(r0, r1);
            const ExactCoef xc = exact_coef_of(r0, r1);
            // green, blue: requested with the rest of the record (inside the blend branch it was an LDS round trip on every blending
            // strip's dependency chain: 0.288 -> 0.2815 ms)
            float2 gb, gb2 = make_float2(0.f, 0.f);
            if (DUAL) { const float4 q2 = lds[3 * j + 2]; gb = make_float2(q2.x, q2.y); gb2 = make_float2(q2.z, q2.w); }  // green, blue of both sets
            else gb = *reinterpret_cast<const float2*>(&lds[3 * j + 2]);
            const uint32_t pos = (uint32_t)(pos_begin + base + j + 1);
            uint64_t stopped_any = 0ull;
            WG_CNT(0, 1);
#pragma unroll
            for (int s = 0; s < 4; s++) {
                if (((reach[s] >> j) & 1ull) == 0ull) continue;  // wave-uniform
                WG_CNT(1, 1);
                WG_CNT(2, __popcll(alive_m[s]));
                // The decisions as wave-wide masks built from the compares' own results (a ballot of a DIRECT compare is the compare's scalar
                // output; combining them is scalar arithmetic): no vector instruction is spent on a predicate.
                float alpha;
                uint64_t pass_m;
                if (EXACT) {
                    float dx, dy, G;
                    const float power = eval_alpha_exact_values(xc, pfx[s], pfy[s], dx, dy, G, alpha);
                    pass_m = __builtin_amdgcn_ballot_w64(!(power > 0.0f)) & __builtin_amdgcn_ballot_w64(!(alpha < (1.0f / 255.0f)));
                } else {
                    PairEval e;
                    const float p2 = eval_alpha_values(sc, pfx[s], pfy[s], e);
                    alpha = e.alpha;
                    pass_m = __builtin_amdgcn_ballot_w64(!(p2 > 0.0f)) & __builtin_amdgcn_ballot_w64(!(alpha < (1.0f / 255.0f)));
                }
                const uint64_t go_m = pass_m & alive_m[s];
                WG_CNT(3, __popcll(go_m));
                WG_CNT(5, go_m == 0ull ? 1 : 0);
                const float w = alpha * T[s];
                // T (1 - alpha), forward.cu:367: as spelled there (EXACT), or one rounding step apart
                const float test_T = EXACT ? ref_test_T(T[s], alpha) : T[s] - w;
                const uint64_t stop_m = go_m & __builtin_amdgcn_ballot_w64(test_T < 0.0001f);   // done (forward.cu:368-372): this instance is not blended
                if (__builtin_amdgcn_inverse_ballot_w64(go_m & ~stop_m)) {
                    Cr[s] += r1.w * w;
                    Cg[s] += gb.x * w;
                    Cb[s] += gb.y * w;
                    if (DUAL) {
                        C2r[s] += r1.z * w;
                        C2g[s] += gb2.x * w;
                        C2b[s] += gb2.y * w;
                    }
                    T[s] = test_T;
                    last[s] = pos;
                }
                alive_m[s] &= ~stop_m;
                stopped_any |= stop_m;
            }
            if (stopped_any != 0ull) {  // some pixel saturated: refresh the strip liveness (scalar)
                strips_alive = 0;
#pragma unroll
                for (int s = 0; s < 4; s++) {
                    if (alive_m[s] != 0ull) strips_alive |= 1u << s;
                    else reach[s] = 0ull;
                }
                if (strips_alive == 0) break;
                todo &= reach[0] | reach[1] | reach[2] | reach[3];  // instances only dead strips could see are dropped
            }
        }
    }
    uint32_t alive = 0;
#pragma unroll
    for (int s = 0; s < 4; s++) alive |= (uint32_t)((alive_m[s] >> lane) & 1ull) << s;
#if WG_COUNT_PAIRS
#pragma unroll
    for (int s = 0; s < 4; s++) WG_CNT(4, __popcll(__ballot((st.alive >> s) & 1u)) - __popcll(__ballot((alive >> s) & 1u)));
    if (lane == 0)
        for (int i = 0; i < 6; i++) atomicAdd(&g_fwd_counters[i], wgc[i]);
#endif
#pragma unroll
    for (int s = 0; s < 4; s++) {
        st.T[s] = T[s]; st.Cr[s] = Cr[s]; st.Cg[s] = Cg[s]; st.Cb[s] = Cb[s]; st.last[s] = last[s];
        if (DUAL) { st.C2r[s] = C2r[s]; st.C2g[s] = C2g[s]; st.C2b[s] = C2b[s]; }
    }
    st.alive = alive;
    st.strips_alive = strips_alive;
}

// complete: the tile is finished (every pixel stopped, or its list is exhausted) -> final outputs.  Otherwise the state is
// parked in the same buffers for a later fwd_init(resume): colour without background, -T for stopped pixels.
// Returns the tile's walked length so far (max of its pixels' last contributors; wave-uniform).
template <bool DUAL = false>
__device__ uint32_t fwd_store(const FwdTile& st, bool complete, int W, int H, int tile, int lane, const float* __restrict__ bg,
                          float* __restrict__ final_T, uint32_t* __restrict__ n_contrib, uint32_t* __restrict__ tile_last,
                          float* __restrict__ out_color, float* __restrict__ accum, float* __restrict__ out_color2 = nullptr,
                          bool colour_only = false) {
    // colour_only (the replay of geometry reuse): the per-pixel state belongs to the parent call, whose `accumulation` view the caller
    // may be holding, and this pass recomputed the identical values: only the image is written
    uint32_t lmax = 0;
    const size_t plane = (size_t)W * H;
    const float bg0 = complete ? bg[0] : 0.f, bg1 = complete ? bg[1] : 0.f, bg2 = complete ? bg[2] : 0.f;
#pragma unroll
    for (int s = 0; s < 4; s++) {
        const int px = st.x0 + strip_x(lane, s), py = st.y0 + strip_y(lane, s);
        if (px < W && py < H) {
            const size_t pix = (size_t)W * py + px;
            if (!colour_only) {
                final_T[pix] = (complete || ((st.alive >> s) & 1u)) ? st.T[s] : -st.T[s];
                if (complete) accum[pix] = 1.0f - st.T[s];  // accumulation (__init__.py:101-113 of the reference computes it from final_T)
                n_contrib[pix] = st.last[s];
            }
            out_color[pix] = st.Cr[s] + st.T[s] * bg0;
            out_color[plane + pix] = st.Cg[s] + st.T[s] * bg1;
            out_color[2 * plane + pix] = st.Cb[s] + st.T[s] * bg2;
            if (DUAL) {
                out_color2[pix] = st.C2r[s] + st.T[s] * bg0;
                out_color2[plane + pix] = st.C2g[s] + st.T[s] * bg1;
                out_color2[2 * plane + pix] = st.C2b[s] + st.T[s] * bg2;
            }
            lmax = max(lmax, st.last[s]);
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) lmax = max(lmax, (uint32_t)__shfl_xor((int)lmax, m));
    if (lane == 0 && !colour_only) tile_last[tile] = lmax;
    return lmax;
}

// seg_end == nullptr: the whole list is sorted (default).  Otherwise (lazy sort, binning.hip) only the first seg_end[tile]
// entries are; a tile that is still accumulating when they run out parks its state and reports the length it consumed in
// tile_state[tile] (0xffffffff = finished) for the fix-up kernel.
// The exact strip test's staging temporaries would take the kernel from 62 to 67 VGPRs, i.e. from 8 to 7 waves per SIMD; the walk wants
// all 8 (it is bounded by dependency chains that only other waves can fill), so the allocation is pinned to 64 and two
// values that are only needed after the walk (the tile's store addresses) live in scratch across it.
#ifndef WG_FWD_WAVES
#define WG_FWD_WAVES 8
#endif
#if WG_FWD_WAVES
#define WG_FWD_OCC __attribute__((amdgpu_waves_per_eu(WG_FWD_WAVES, WG_FWD_WAVES)))
#else
#define WG_FWD_OCC
#endif
// WG_PROBE (a VARIANT build only: scripts/probe_balance.py): every wave records its start / end on the 100 MHz real-time counter, the SIMD it
// ran on (HW_ID, XCC_ID) and its tile -- how evenly the launch order loads the 1 024 SIMDs.
#ifndef WG_PROBE
#define WG_PROBE 0
#endif
#if WG_PROBE
__device__ unsigned long long g_fwd_probe[4 * 65536];
#endif
template <bool EXACT, bool DUAL>
__device__ __forceinline__ void render_forward_body(
    float4* lds, int W, int H, int gx, int tiles, const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
    const float4* __restrict__ splats, const float2* __restrict__ subpixel_offset, const float* __restrict__ bg,
    const uint32_t* seg_end, uint32_t* __restrict__ tile_state,
    float* __restrict__ final_T, uint32_t* __restrict__ n_contrib, uint32_t* tile_last,
    float* __restrict__ out_color, const BinStats* __restrict__ guard, int replay, float* __restrict__ accum, float* __restrict__ out_color2,
    const uint32_t* __restrict__ order, const uint32_t* __restrict__ order_key, uint32_t* __restrict__ order_table, uint32_t order_slots,
    uint32_t order_stride) {
    if (guard && guard->spec_fail) return;  // speculative forward (api.hip): the frame did not fit what was enqueued; the host re-issues
#if WG_PROBE
    const unsigned long long probe_t0 = __builtin_amdgcn_s_memrealtime();
#endif
    // order (binning.hip: forward_order_kernel): a permutation of each XCD band's tiles by the costs an earlier frame of this camera had
    const int tile = order ? (int)order[xcd_tile(blockIdx.x, tiles)] : xcd_tile(blockIdx.x, tiles);
    const int lane = threadIdx.x;
    FwdTile st;
    fwd_init<DUAL>(st, W, H, gx, tile, lane, subpixel_offset, false, final_T, n_contrib, out_color, out_color2);
    const uint2 range = ranges[tile];
    const int n = (int)(range.y - range.x);
    const int end = seg_end ? min(n, (int)seg_end[tile]) : n;
    fwd_walk<true, EXACT, DUAL>(st, lds, lane, point_list + range.x, splats, 0, end);
    // replay (geometry reuse): seg_end is the tile's walked length of an earlier pass over the same geometry -- nothing behind it
    // contributes, so the tile is complete whatever is left of the list
    const bool complete = st.strips_alive == 0 || end == n || replay != 0;
    const uint32_t walked = fwd_store<DUAL>(st, complete, W, H, tile, lane, bg, final_T, n_contrib, tile_last, out_color, accum, out_color2, replay != 0);
    if (tile_state && lane == 0) tile_state[tile] = complete ? 0xffffffffu : (uint32_t)end;
    if (order_table && lane == 0) {   // this frame's cost of the tile, for the next frame of the camera (a hint: no ordering, no atomics)
        const uint32_t slot = order_key[0];
        order_table[2 * (size_t)order_slots + (size_t)slot * order_stride + tile] = walked + 1u;   // (+1: staging and stores cost something for an empty tile too)
        if (blockIdx.x == 0) { order_table[2 * (size_t)slot] = order_key[1]; order_table[2 * (size_t)slot + 1] = order_key[2]; }
    }
#if WG_PROBE
    if (lane == 0 && blockIdx.x < 65536) {
        g_fwd_probe[4 * blockIdx.x] = probe_t0;
        g_fwd_probe[4 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
        g_fwd_probe[4 * blockIdx.x + 2] = ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32) |   // XCC_ID
                                          (unsigned)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));                      // HW_ID
        g_fwd_probe[4 * blockIdx.x + 3] = (unsigned long long)tile;
    }
#endif
}

// (seg_end and tile_last are the same array in the replay launch: neither is __restrict__)
template <bool EXACT>
__global__ void __launch_bounds__(64) WG_FWD_OCC render_forward_kernel(
    int W, int H, int gx, int tiles, const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
    const float4* __restrict__ splats, const float2* __restrict__ subpixel_offset, const float* __restrict__ bg,
    const uint32_t* seg_end, uint32_t* __restrict__ tile_state,
    float* __restrict__ final_T, uint32_t* __restrict__ n_contrib, uint32_t* tile_last,
    float* __restrict__ out_color, const BinStats* __restrict__ guard, int replay, float* __restrict__ accum,
    const uint32_t* __restrict__ order, const uint32_t* __restrict__ order_key, uint32_t* __restrict__ order_table, uint32_t order_slots, uint32_t order_stride) {
    __shared__ float4 lds[BATCH * 3];
    render_forward_body<EXACT, false>(lds, W, H, gx, tiles, ranges, point_list, splats, subpixel_offset, bg, seg_end, tile_state, final_T, n_contrib,
                                      tile_last, out_color, guard, replay, accum, nullptr, order, order_key, order_table, order_slots, order_stride);
}

// the two-colour walk keeps twelve more sums per lane: its register allocation is left to the compiler (no occupancy pin)
#ifndef WG_FWD_DUAL_WAVES
#define WG_FWD_DUAL_WAVES 6   // 80 VGPRs (two spilled outside the loop): 1.737 / 1.711 ms against 1.743 / 1.751 unpinned (3 M real-caller replay, profiles/r4/ab_two_colour_occupancy.txt)
#endif
#if WG_FWD_DUAL_WAVES
#define WG_FWD_DUAL_OCC __attribute__((amdgpu_waves_per_eu(WG_FWD_DUAL_WAVES, WG_FWD_DUAL_WAVES)))
#else
#define WG_FWD_DUAL_OCC
#endif
template <bool EXACT>
__global__ void __launch_bounds__(64) WG_FWD_DUAL_OCC render_forward_dual_kernel(
    int W, int H, int gx, int tiles, const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
    const float4* __restrict__ splats, const float2* __restrict__ subpixel_offset, const float* __restrict__ bg,
    const uint32_t* seg_end, uint32_t* __restrict__ tile_state,
    float* __restrict__ final_T, uint32_t* __restrict__ n_contrib, uint32_t* tile_last,
    float* __restrict__ out_color, const BinStats* __restrict__ guard, float* __restrict__ accum, float* __restrict__ out_color2,
    const uint32_t* __restrict__ order, const uint32_t* __restrict__ order_key, uint32_t* __restrict__ order_table, uint32_t order_slots, uint32_t order_stride) {
    __shared__ float4 lds[BATCH * 3];
    render_forward_body<EXACT, true>(lds, W, H, gx, tiles, ranges, point_list, splats, subpixel_offset, bg, seg_end, tile_state, final_T, n_contrib,
                                     tile_last, out_color, guard, 0, accum, out_color2, order, order_key, order_table, order_slots, order_stride);
}

hipError_t launch_render_forward(int W, int H, int gx, int gy, const ImageState& img, const BinningState& b,
                                 const GeometryState& g, const float* subpixel_offset, const float* background,
                                 float* out_color, float* out_color2, bool lazy, bool exact, const BinStats* guard, uint32_t* order_table,
                                 uint32_t order_slots, uint32_t order_stride, hipStream_t stream) {
    const int tiles = gx * gy;
    if (tiles <= 0) return hipSuccess;
    const uint32_t* order = order_table ? img.order_fwd : (const uint32_t*)nullptr;
    const uint32_t* okey = order_table ? img.order_key : (const uint32_t*)nullptr;
#define WG_LAUNCH(EX)                                                                                                                       \
    hipLaunchKernelGGL(render_forward_kernel<EX>, dim3(tiles), dim3(64), 0, stream, W, H, gx, tiles, img.ranges, b.point_list, g.splats,   \
                       reinterpret_cast<const float2*>(subpixel_offset), background, lazy ? img.seg_end : (const uint32_t*)nullptr,        \
                       lazy ? img.tile_state : (uint32_t*)nullptr, img.final_T, img.n_contrib, img.tile_last, out_color, guard, 0, img.accum, \
                       order, okey, order_table, order_slots, order_stride)
#define WG_LAUNCH_DUAL(EX)                                                                                                                  \
    hipLaunchKernelGGL(render_forward_dual_kernel<EX>, dim3(tiles), dim3(64), 0, stream, W, H, gx, tiles, img.ranges, b.point_list, g.splats, \
                       reinterpret_cast<const float2*>(subpixel_offset), background, lazy ? img.seg_end : (const uint32_t*)nullptr,        \
                       lazy ? img.tile_state : (uint32_t*)nullptr, img.final_T, img.n_contrib, img.tile_last, out_color, guard, img.accum, out_color2, \
                       order, okey, order_table, order_slots, order_stride)
    if (out_color2) { if (exact) WG_LAUNCH_DUAL(true); else WG_LAUNCH_DUAL(false); }
    else { if (exact) WG_LAUNCH(true); else WG_LAUNCH(false); }
#undef WG_LAUNCH
#undef WG_LAUNCH_DUAL
    return hipGetLastError();
}

hipError_t launch_render_forward_replay(int W, int H, int gx, int gy, const ImageState& img, const BinningState& b,
                                        const GeometryState& g, const float* subpixel_offset, const float* background,
                                        float* out_color, bool exact, hipStream_t stream) {
    const int tiles = gx * gy;
    if (tiles <= 0) return hipSuccess;
#define WG_LAUNCH(EX)                                                                                                                       \
    hipLaunchKernelGGL(render_forward_kernel<EX>, dim3(tiles), dim3(64), 0, stream, W, H, gx, tiles, img.ranges, b.point_list, g.splats,   \
                       reinterpret_cast<const float2*>(subpixel_offset), background, (const uint32_t*)img.tile_last, (uint32_t*)nullptr,   \
                       img.final_T, img.n_contrib, img.tile_last, out_color, (const BinStats*)nullptr, 1, img.accum,                        \
                       (const uint32_t*)nullptr, (const uint32_t*)nullptr, (uint32_t*)nullptr, 0u, 0u)
    if (exact) WG_LAUNCH(true); else WG_LAUNCH(false);
#undef WG_LAUNCH
    return hipGetLastError();
}

__global__ void __launch_bounds__(256) poison_unfit_kernel(const BinStats* __restrict__ stats, size_t N, int tiles, float* __restrict__ out_color,
                                                           float* __restrict__ accum, uint32_t* __restrict__ n_contrib, uint32_t* __restrict__ tile_last,
                                                           float* __restrict__ out_color2) {
    if (stats->spec_fail == 0u) return;  // the usual case: one scalar load per workgroup
    const float nan = __builtin_nanf("");
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (size_t)gridDim.x * blockDim.x) {
        out_color[i] = nan; out_color[N + i] = nan; out_color[2 * N + i] = nan;
        if (out_color2) { out_color2[i] = nan; out_color2[N + i] = nan; out_color2[2 * N + i] = nan; }
        accum[i] = nan;
        n_contrib[i] = 0u;
        if (i < (size_t)tiles) tile_last[i] = 0u;
    }
}

hipError_t launch_poison_unfit(const ImageState& img, int W, int H, int tiles, float* out_color, float* out_color2, hipStream_t stream) {
    const size_t N = (size_t)W * H;   // (tiles <= N always: a tile holds at least one pixel)
    hipLaunchKernelGGL(poison_unfit_kernel, dim3(256), dim3(256), 0, stream, img.stats, N, tiles, out_color, img.accum, img.n_contrib, img.tile_last, out_color2);
    return hipGetLastError();
}

// ---- lazy sort, later rounds (fix-up) ---------------------------------------------------------------------------------------
// One 256-thread workgroup per tile, a no-op for finished tiles.  For a tile whose sorted front ran out while pixels were still
// accumulating it loops: split the next front off the unsorted bag (or take all of it when short), sort it in place behind the
// part already consumed, let wave 0 resume the walk over it -- until every pixel has stopped or the list is exhausted.  The
// final list prefix is in exactly the order a full sort gives, so n_contrib / tile_last / the backward pass are unaffected.
template <bool EXACT, bool DUAL>
__global__ void __launch_bounds__(256) render_fixup_kernel(
    int W, int H, int gx, int tiles, const uint2* __restrict__ ranges, uint32_t* point_list, const uint32_t* __restrict__ bucket_ids,
    const float* __restrict__ depths, const float4* __restrict__ splats, const float2* __restrict__ subpixel_offset,
    const float* __restrict__ bg, uint32_t* __restrict__ tile_state, float* final_T, uint32_t* n_contrib,
    uint32_t* __restrict__ tile_last, float* out_color, uint32_t target, uint32_t cap, uint32_t id_mask,
    const uint32_t* __restrict__ tile_near, SplitState* split, int phase, HostMailbox* mailbox, const BinStats* __restrict__ guard,
    float* __restrict__ accum, float* out_color2) {
    __shared__ float4 lds[BATCH * 3];
    __shared__ uint64_t skeys[256 * 8];
    __shared__ SelectScratch sc;
    __shared__ int s_complete;
    if (guard && guard->spec_fail) return;
    // phase 1 runs after every tile's phase 0 (stream order): which bands asked for far instances is final, and goes to the host's
    // mailbox as the hint for later frames (api.hip: a thread whose frames keep needing the far phase stops attempting the split)
    if (phase == 1 && mailbox && blockIdx.x == 0 && threadIdx.x == 0)
        __hip_atomic_store(&mailbox->far_report,
                           ((unsigned long long)split->aim << 32) | (1u + min(split->far_tiles, 0xffffffu) + ((uint32_t)__popc(split->need_far & 0xffu) << 24)),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // tiles that asked (24 bits), in how many of the eight bands (what the far scatter costs), the frame's aim
    const int tile = xcd_tile(blockIdx.x, tiles);
    uint32_t done = tile_state[tile];
    if (done == 0xffffffffu) return;  // workgroup-uniform
    const int tid = threadIdx.x;
    const bool walker = tid < 64;
    const uint2 range = ranges[tile];
    const uint32_t n = range.y - range.x;
    // Near / far split (binning.hip): the bucket holds the near instances in [0, n_near) -- all there is in phase 0 -- and, once the
    // far scatter has run for this tile, the far ones in [n_near, n).  Without a split n_near == n and phase 1 never runs.
    const uint32_t n_near = tile_near ? tile_near[tile] : n;
    const uint32_t bag_lo = phase == 0 ? 0u : n_near, bag_hi = phase == 0 ? n_near : n;
    if (done >= bag_hi) {  // this bag is used up already (phase 0: the front was all of the near instances, or there are none)
        if (phase == 0 && tid == 0) {  // n_near < n here: a finished tile does not arrive
            atomicOr(&split->need_far, 1u << (blockIdx.x & 7));
            atomicAdd(&split->far_tiles, 1u);
        }
        return;
    }
    const uint32_t* bag = bucket_ids + range.x + bag_lo;
    uint32_t* list = point_list + range.x;
    FwdTile st;
    if (walker) fwd_init<DUAL>(st, W, H, gx, tile, tid, subpixel_offset, true, final_T, n_contrib, out_color, out_color2);
    bool complete = false;
    for (;;) {
        // everything at or below the last sorted key has been taken (nothing yet when the tile had no sorted front at all)
        const uint64_t lo = done > 0 ? depth_key(depths, list[done - 1]) : 0ull;
        const uint32_t F = extract_front(bag, bag_hi - bag_lo, depths, lo, bag_hi - done, target, cap, id_mask, sc);
        if (F <= 1024) tile_sort_body<4>(skeys, F, sc.ids, depths, list + done, 0xffffffffu);
        else tile_sort_body<8>(skeys, F, sc.ids, depths, list + done, 0xffffffffu);
        __threadfence_block();
        __syncthreads();  // the sorted segment is visible to wave 0
        if (walker) {
            fwd_walk<false, EXACT, DUAL>(st, lds, tid, list, splats, (int)done, (int)(done + F));
            if (tid == 0) s_complete = (st.strips_alive == 0 || done + F == n) ? 1 : 0;
        }
        done += F;
        __syncthreads();
        complete = s_complete != 0;
        if (complete || done >= bag_hi) break;
        __syncthreads();  // s_complete is rewritten next round
    }
    // complete: final outputs.  Otherwise the near bag ran out with pixels still accumulating: park the state again and ask for
    // the far instances (phase 1 continues from here).
    if (walker) fwd_store<DUAL>(st, complete, W, H, tile, tid, bg, final_T, n_contrib, tile_last, out_color, accum, out_color2);
    if (tid == 0) {
        tile_state[tile] = complete ? 0xffffffffu : done;
        if (!complete) {
            atomicOr(&split->need_far, 1u << (blockIdx.x & 7));  // xcd_tile: block b serves a tile of band b & 7
            atomicAdd(&split->far_tiles, 1u);
        }
    }
}

hipError_t launch_render_fixup(int code_bits, int W, int H, int gx, int gy, const ImageState& img, const BinningState& b, const GeometryState& g,
                               const float* subpixel_offset, const float* background, float* out_color, float* out_color2, const LazyConfig& g_lazy,
                               bool split, int phase, bool exact, HostMailbox* mailbox_dev, const BinStats* guard, hipStream_t stream) {
    const int tiles = gx * gy;
    if (tiles <= 0) return hipSuccess;
#define WG_LAUNCH(EX, DU)                                                                                                                   \
    hipLaunchKernelGGL((render_fixup_kernel<EX, DU>), dim3(tiles), dim3(256), 0, stream, W, H, gx, tiles, img.ranges, b.point_list, b.bucket_ids, \
                       g.depths, g.splats, reinterpret_cast<const float2*>(subpixel_offset), background, img.tile_state, img.final_T,      \
                       img.n_contrib, img.tile_last, out_color, 1536u < g_lazy.cap ? 1536u : (g_lazy.cap * 3u) / 4u,                       \
                       g_lazy.cap < FRONT_CAP ? g_lazy.cap : FRONT_CAP, code_bits ? (1u << (32 - code_bits)) - 1u : 0xffffffffu,           \
                       split ? img.tile_near : (const uint32_t*)nullptr, img.split, phase, mailbox_dev, guard, img.accum, out_color2)
    if (out_color2) { if (exact) WG_LAUNCH(true, true); else WG_LAUNCH(false, true); }
    else { if (exact) WG_LAUNCH(true, false); else WG_LAUNCH(false, false); }
#undef WG_LAUNCH
    return hipGetLastError();
}

}  // namespace wg

#if WG_PROBE
extern "C" int wg_probe_fetch(void* dst, size_t bytes) {
    hipError_t e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpyFromSymbol(dst, HIP_SYMBOL(wg::g_fwd_probe), bytes < sizeof(wg::g_fwd_probe) ? bytes : sizeof(wg::g_fwd_probe), 0, hipMemcpyDeviceToHost);
    return (int)e;
}
#endif
#if WG_COUNT_PAIRS
extern "C" int wg_debug_fwd_counters(unsigned long long* out8, int reset) {
    hipError_t e = hipDeviceSynchronize();
    if (e == hipSuccess && out8) e = hipMemcpyFromSymbol(out8, HIP_SYMBOL(wg::g_fwd_counters), 8 * sizeof(unsigned long long));
    if (e == hipSuccess && reset) {
        const unsigned long long z[8] = {};
        e = hipMemcpyToSymbol(HIP_SYMBOL(wg::g_fwd_counters), z, sizeof(z));
    }
    return e == hipSuccess ? 0 : -3;
}
#endif
