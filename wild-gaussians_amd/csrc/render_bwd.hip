// K9: per-tile back-to-front gradient walk for gfx950.  Replaces renderCUDA (backward.cu:435-606).
//
// Same wave-per-tile mapping as the forward kernel (render_fwd.hip): 64 lanes x 4 strips = 256 pixels, the same
// batch staging through LDS and the same per-instance strip masks.  What changes relative to the reference:
//   * the walk starts at the tile's last contributing instance (tile_last, produced by the forward kernel)
//     instead of the end of the tile's list, so instances no pixel ever reached are never read;
//   * the reference issues 10 float atomicAdds to global memory per contributing (pixel, Gaussian) pair
//     (backward.cu:568-603).  Here each lane first sums its 4 pixels' contributions in registers, then the wave
//     reduces the 10 partial sums with a butterfly (transpose-reduce): v_permlane32_swap / v_permlane16_swap
//     halve the number of live values per step instead of reducing them one by one (26 VALU ops for all ten,
//     against 60 for ten 6-step DPP reductions), leaving value k's total in lane LANE_OF[k].  Those ten lanes
//     then issue ONE global_atomic_add_f32 instruction per (tile, Gaussian) instance: 256x fewer atomics than the
//     reference for a fully covered tile;
//   * the "last_alpha / last_color" lazy recurrence (backward.cu:560-561,572) is applied eagerly
//     (accum_rec' = alpha*c + (1-alpha)*accum_rec right after use) and accum_rec is carried as its dot product with the
//     pixel's dL_dpixel (the only way it is ever used): one register per pixel instead of seven.
// The gradient arithmetic is the reference's hand-derived backward, not autograd of the forward:
// straight-through min(0.99,.), NDC-scaled dL_dmean2D (0.5*W, 0.5*H), abs-gradient in .z
// (backward.cu:593-595), conic gradient in .x/.y/.w of a float4 (backward.cu:598-600).
#include <algorithm>
#include "wg_common.h"
#include "wg_alpha.h"

namespace wg {

constexpr int BATCH = 64;

// WG_COUNT_PAIRS (a VARIANT build only, see render_fwd.hip): [0] instances visited, [1] strip evaluations (one = 64 (pixel, entry) pairs),
// [2] of those pairs, the ones at or before their pixel's last contributor, [3] contributing pairs (what backward.cu:536-600
// differentiates), [4] instances reduced and added to their Gaussian's record, [5] strip evaluations in which no pair contributed,
// [6..9] reduced instances by the number of LANES that contributed anything (1, 2-4, 5-16, 17-64: what a path without the butterfly would
// have to serve), [10..13] strip evaluations by contributing lanes (1-8, 9-24, 25-48, 49-64).
#ifndef WG_COUNT_PAIRS
#define WG_COUNT_PAIRS 0
#endif
#if WG_COUNT_PAIRS
__device__ unsigned long long g_bwd_counters[16];
#define WG_CNT(i, v) wgc[i] += (unsigned long long)(v)
#else
#define WG_CNT(i, v)
#endif

// Measured and rejected variants of this kernel (packed f32 per-pair math, scalar any-mask, two instances per butterfly, the reduction
// on the matrix pipe, forced occupancy) are NOT in this file: experiments/r3_render_bwd_variants.patch adds them back as compile-time
// knobs for scripts/ab_variants.sh; their numbers are in EXPERIMENTS.md (R3.1).
typedef float f2 __attribute__((ext_vector_type(2)));

template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}

// pair_x32: (a, b) -> one register: lanes 0-31 hold a[l] + a[l+32], lanes 32-63 hold b[l-32] + b[l]
// pair_x16: (a, b) -> one register: even 16-lane rows hold a summed over the row pair, odd rows hold b likewise
__device__ __forceinline__ float pair_x32(float a, float b) {
    // hipcc 7.2 mis-selects "r[0] + r[1]" of __builtin_amdgcn_permlane32_swap as "r[0] + r[0]"; passing the two results through an
    // empty asm keeps them apart.  (Spelling the swap itself in asm works too but needs hand-placed s_nops for the
    // VALU-write -> swap -> VALU-read wait states, which the compiler otherwise fills with useful instructions.)
    auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
    unsigned x = r[0], y = r[1];
    asm volatile("" : "+v"(x), "+v"(y));
    return __builtin_bit_cast(float, x) + __builtin_bit_cast(float, y);
}
__device__ __forceinline__ float pair_x16(float a, float b) {
    auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
    unsigned x = r[0], y = r[1];
    asm volatile("" : "+v"(x), "+v"(y));
    return __builtin_bit_cast(float, x) + __builtin_bit_cast(float, y);
}

// Butterfly reduction of ten per-lane values over the 64 lanes of a wave.  On return lane l holds, in the
// returned register, the full-wave total of value index
//     bit1(l) == 0 :  4*bit0(l) + 2*bit4(l) + bit5(l)        (values 0..7)
//     bit1(l) == 1 :  8 + bit5(l)                             (values 8, 9)
// (bits 2 and 3 of the lane index do not matter: those lanes hold copies).
__device__ __forceinline__ float butterfly10(float v0, float v1, float v2, float v3, float v4, float v5, float v6, float v7,
                                             float v8, float v9, int lane) {
    // xor 32: ten -> five;  w_k = v_{2k + bit5}
    const float w0 = pair_x32(v0, v1), w1 = pair_x32(v2, v3), w2 = pair_x32(v4, v5), w3 = pair_x32(v6, v7), w4 = pair_x32(v8, v9);
    // xor 16: five -> three;  u_m = w_{2m + bit4}, u2 = w4
    const float u0 = pair_x16(w0, w1), u1 = pair_x16(w2, w3), u2 = pair_x16(w4, w4);
    // xor 1 (quad_perm [1,0,3,2]): x0 = u_{bit0}, x1 = u2
    const bool b0 = lane & 1;
    const float keep0 = b0 ? u1 : u0, send0 = b0 ? u0 : u1;
    const float x0 = keep0 + dpp_f<0xB1>(send0);
    const float x1 = u2 + dpp_f<0xB1>(u2);
    // xor 2 (quad_perm [2,3,0,1]): y = bit1 ? x1 : x0
    const bool b1 = lane & 2;
    const float keep1 = b1 ? x1 : x0, send1 = b1 ? x0 : x1;
    float y = keep1 + dpp_f<0x4E>(send1);
    // lanes l, l+4, l+8, l+12 of each row: row_ror 4, row_ror 8
    y += dpp_f<0x124>(y);
    y += dpp_f<0x128>(y);
    return y;
}


// The same for up to sixteen values (the two-colour walk reduces thirteen): on return lane l holds the total of value index
//     8*bit1(l) + 4*bit0(l) + 2*bit4(l) + bit5(l)
// (bits 2 and 3 of the lane index do not matter).  Absent values are passed as 0.f: their pair adds fold away.
__device__ __forceinline__ float butterfly13(float v0, float v1, float v2, float v3, float v4, float v5, float v6, float v7, float v8, float v9,
                                             float v10, float v11, float v12, int lane) {
    // xor 32: w_k = v_{2k + bit5}
    const float w0 = pair_x32(v0, v1), w1 = pair_x32(v2, v3), w2 = pair_x32(v4, v5), w3 = pair_x32(v6, v7), w4 = pair_x32(v8, v9),
                w5 = pair_x32(v10, v11), w6 = pair_x32(v12, v12);
    // xor 16: u_m = w_{2m + bit4}
    const float u0 = pair_x16(w0, w1), u1 = pair_x16(w2, w3), u2 = pair_x16(w4, w5), u3 = pair_x16(w6, w6);
    // xor 1: x_n = u_{2n + bit0}
    const bool b0 = lane & 1;
    const float x0 = (b0 ? u1 : u0) + dpp_f<0xB1>(b0 ? u0 : u1);
    const float x1 = (b0 ? u3 : u2) + dpp_f<0xB1>(b0 ? u2 : u3);
    // xor 2: y = x_{bit1}
    const bool b1 = lane & 2;
    float y = (b1 ? x1 : x0) + dpp_f<0x4E>(b1 ? x0 : x1);
    y += dpp_f<0x124>(y);
    y += dpp_f<0x128>(y);
    return y;
}

// RECORD (default, wg_set_option("grad_record")): the ten reduced values of an instance go, unscaled, to ONE 48-byte gradient
// record of its Gaussian (grad_rec[12 id + k], k = the value's index; wg_common.h: GRAD_REC_*) -- one L2 line per instance (two for
// a quarter of the records) instead of partial lines of four arrays, and the per-Gaussian factors (opacity, 0.5 W,
// -0.5, 1 / log2 e) are applied once per Gaussian by the preprocess backward kernel, which also writes the caller-visible
// dL_dmean2D / dL_dconic / dL_dopacity / dL_dcolor.  !RECORD: the arrays themselves are the accumulation targets (as in the
// reference, backward.cu:568-603), factors applied per instance.
// DET (wg_set_option("deterministic_backward"), implies RECORD): no atomics at all.  Every (tile, Gaussian) instance owns a
// slot of ten floats -- slot = (exclusive prefix of tiles_touched)[id] + the tile's index inside the Gaussian's tile rectangle --
// into which the ten lanes STORE the wave-reduced sums; det_reduce_kernel then adds a Gaussian's slots in slot order into its
// gradient record.  Same sums as the atomic path up to the order of a Gaussian's per-tile terms, which is now fixed: two runs
// give bit-identical gradients.
// EXACT (Options::exact_compositing): the forward kernel took every skip decision on the reference's own arithmetic (wg_alpha.h:
// eval_alpha_exact).  This kernel keeps the fast evaluation (its values only have to be accurate) but must take the SAME decisions, so
// a pair whose fast values are within a proven error band of a threshold is re-evaluated with that arithmetic (a rare,
// wave-uniform branch; ~10^3 pairs of ~10^9 per frame) and its decision, alpha and G are taken from there.
#ifndef WG_PROBE
#define WG_PROBE 0   // a VARIANT build only (render_fwd.hip; scripts/probe_balance.py)
#endif
#if WG_PROBE
__device__ unsigned long long g_bwd_probe[4 * 65536];
#endif
#ifndef WG_BWD_WAVES
#define WG_BWD_WAVES 0
#endif
#if WG_BWD_WAVES
#define WG_BWD_OCC __attribute__((amdgpu_waves_per_eu(WG_BWD_WAVES, WG_BWD_WAVES)))
#else
#define WG_BWD_OCC
#endif
// DUAL (RECORD): two colour sets over one walk (wg_second_image, include/wg_rasterizer.h).  The record's spare floats carry the second
// colour; each set keeps its own dL_dalpha chain (accum_rec, background term), their sum feeds the nine geometry sums -- linear in it --
// and the abs-gradient takes |q1| + |q2|, as two calls would accumulate it; thirteen sums are reduced per instance instead of 2 x 10.  Sums 10, 11 go to the record's two spare floats, sum 12 to grad_aux[id].
#ifndef WG_BWD_DUAL_WAVES
#define WG_BWD_DUAL_WAVES 0
#endif
// WG_BWD_LDS_REDUCE (default path only: RECORD, not DET, not DUAL): the ten per-lane sums of an instance are reduced through LDS behind the
// butterfly's FIRST stage instead of through all of it.  The xor-32 stage leaves five sums per lane (lanes 0 - 31: values 0, 2, 4, 6, 8,
// lanes 32 - 63: values 1, 3, 5, 7, 9); every lane parks them as a row of five floats (five ds_write_b32, row stride 20 bytes: 5 is coprime
// with the 64 banks, no conflict); lane 32 half + 4 m + h then adds column m of its half's rows h, h + 4, ... h + 28 (four ds_read2_b32 -- at
// a fixed step the forty reading lanes touch forty different banks, lanes with m >= 5 re-read columns 0 - 2, a broadcast) and two
// quad-permute adds complete the column: 10 + 7 + 2 vector instructions where the rest of the butterfly spent 3 swaps + 3 adds + 4 selects +
// 5 DPP adds + moves + 9 wait states.  The LDS instructions issue beside the other waves' vector instructions (the kernel is bound by vector
// issue), and one wave's LDS operations execute in program order: no wait between the writes and the reads, none before the next
// instance's writes.  Same box, K9 at the headline frame: butterfly 0.4188, all ten sums through LDS (640 floats per instance written, 1024
// read) 0.4067, this form (320 / 512) 0.3961 ms; with ds_write_b64 rows (stride 24 bytes, halves skewed by 32 banks) 0.3954: the LDS
// traffic is what the variants differ in, not the instruction count (profiles/r6/ab_k9_reduction_through_lds.txt).
#ifndef WG_BWD_LDS_REDUCE
#define WG_BWD_LDS_REDUCE 1
#endif
template <bool RECORD, bool DET = false, bool EXACT = false, bool DUAL = false>
__global__ void __launch_bounds__(64) WG_BWD_OCC
#if WG_BWD_DUAL_WAVES
__attribute__((amdgpu_waves_per_eu(DUAL ? WG_BWD_DUAL_WAVES : 1, DUAL ? WG_BWD_DUAL_WAVES : 8)))
#endif
render_backward_kernel(
    int W, int H, int gx, int tiles, const uint32_t* __restrict__ order, const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
    const float4* __restrict__ splats, const float2* __restrict__ subpixel_offset, const float* __restrict__ bg,
    const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib, const uint32_t* __restrict__ tile_last,
    const float* __restrict__ dL_dpix, float* __restrict__ dL_dmean2D, float* __restrict__ dL_dconic,
    float* __restrict__ dL_dopacity, float* __restrict__ dL_dcolor, float* __restrict__ grad_rec,
    const ushort4* __restrict__ rects, const uint32_t* __restrict__ offsets_incl, const uint32_t* __restrict__ tiles_touched,
    float* __restrict__ det_slots, unsigned char* __restrict__ det_flags, const float* __restrict__ dL_dpix2, float* __restrict__ grad_aux) {
    static_assert(!DUAL || RECORD, "the two-colour walk accumulates into the gradient record (or, DET, into fourteen-float slots)");
    constexpr int SLOT_FLOATS = DUAL ? 14 : 10;   // deterministic mode: floats per (tile, Gaussian) slot (thirteen sums, padded to 8-byte multiples)
    constexpr int RS = DUAL ? 4 : 3;   // float4 per parked record
    constexpr bool LDSRED = WG_BWD_LDS_REDUCE != 0;
    constexpr int NS = DUAL ? 7 : 5;   // sums per lane behind the butterfly's first stage (ten values: 5, the two-colour walk's thirteen: 7)
    // LDSRED: one row of NS sums per lane, at the START of the allocation (ds_read2 offsets are 8-bit element counts from the lane's base
    // address: behind the records every access would first add the array's offset)
    __shared__ float4 smem[(LDSRED ? 16 * NS : 0) + BATCH * RS];
    float4* const lds = smem + (LDSRED ? 16 * NS : 0);
    float* const red = reinterpret_cast<float*>(smem);

#if WG_PROBE
    const unsigned long long probe_t0 = __builtin_amdgcn_s_memrealtime();
#endif
    const int tile = (int)order[xcd_tile(blockIdx.x, tiles)];
    const int hi0 = (int)tile_last[tile];
#if WG_PROBE
    if (hi0 == 0 && threadIdx.x == 0 && blockIdx.x < 65536) {
        g_bwd_probe[4 * blockIdx.x] = probe_t0; g_bwd_probe[4 * blockIdx.x + 1] = probe_t0; g_bwd_probe[4 * blockIdx.x + 3] = (unsigned long long)tile;
        g_bwd_probe[4 * blockIdx.x + 2] = ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32) | (unsigned)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
    }
#endif
    if (hi0 == 0) return;
    const int lane = threadIdx.x;
    const int tx = tile % gx, ty = tile / gx;
    const size_t plane = (size_t)W * H;
    const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
    const float ddelx_dx = 0.5f * W, ddely_dy = 0.5f * H;

    // which of the ten reduced values this lane owns after butterfly10(), and where it accumulates it:
    //   0,1,2 -> dL_dcolor[3id + k]; 3,4,5 -> dL_dmean2D[3id + k-3]; 6,7,8 -> dL_dconic[4id + {0,1,3}]; 9 -> dL_dopacity[id]
    // (LDSRED: lane 32 half + 4 m holds value 2 m + half, m < NS; value 12 of the two-colour walk comes out in both halves)
    const int red_m = (lane & 31) >> 2, red_half = lane >> 5;
    const int vidx = LDSRED ? min(2 * red_m + red_half, DUAL ? 12 : 9)
                   : DUAL ? 8 * ((lane >> 1) & 1) + 4 * (lane & 1) + 2 * ((lane >> 4) & 1) + ((lane >> 5) & 1)
                          : ((lane & 2) ? 8 + ((lane >> 5) & 1) : 4 * (lane & 1) + 2 * ((lane >> 4) & 1) + ((lane >> 5) & 1));
    const bool owner = (lane & 12) == 0;  // lanes with bits 2,3 clear: one lane per value (two spare for value 8/9 copies)
    float* abase;
    uint32_t astride;
    if (DUAL && vidx >= 12) { abase = grad_aux; astride = 1; }   // (vidx 13..15 hold nothing and never issue)
    else if (RECORD) { abase = grad_rec + vidx; astride = GRAD_REC_FLOATS; }
    else if (vidx < 3) { abase = dL_dcolor + vidx; astride = 3; }
    else if (vidx < 6) { abase = dL_dmean2D + (vidx - 3); astride = 3; }
    else if (vidx < 9) { abase = dL_dconic + (vidx == 8 ? 3 : vidx - 6); astride = 4; }
    else { abase = dL_dopacity; astride = 1; }
    // values 8 and 9 (bit1 set) are replicated over bits 0 and 4: let only the bit0 == bit4 == 0 copy issue
    const bool issue = LDSRED ? ((lane & 3) == 0 && red_m < NS && !(DUAL && red_m == 6 && red_half == 1))
                     : DUAL ? (owner && vidx <= 12) : (owner && !((lane & 2) && (lane & 17)));
    // constant factor of this lane's value (see the per-pair sums below); values 3..8 also carry the splat's opacity
    const bool oscale = vidx >= 3 && vidx <= 8;
    constexpr float INV_L = 1.0f / WG_LOG2E;  // u, v above carry a factor -log2(e)
    const float vscale = vidx == 3 ? ddelx_dx * INV_L : vidx == 4 ? ddely_dy * INV_L : vidx == 5 ? INV_L : (vidx >= 6 && vidx <= 8) ? -0.5f : 1.0f;

    float pfx[4], pfy[4], T[4], tfb[4], dLr[4], dLg[4], dLb[4], recd[4], dL2r[4], dL2g[4], dL2b[4], recd2[4], tfb2[4];
    int last[4];
    StripBounds sb;
#pragma unroll
    for (int s = 0; s < 4; s++) {
        const int px = tx * TILE_X + strip_x(lane, s), py = ty * TILE_Y + strip_y(lane, s);
        const bool inside = px < W && py < H;
        float2 off = make_float2(0.f, 0.f);
        T[s] = 0.f; last[s] = 0; dLr[s] = dLg[s] = dLb[s] = 0.f;
        if (DUAL) dL2r[s] = dL2g[s] = dL2b[s] = 0.f;
        if (inside) {
            const size_t pix = (size_t)W * py + px;
            if (subpixel_offset) off = subpixel_offset[pix];
            T[s] = final_T[pix];
            last[s] = (int)n_contrib[pix];
            dLr[s] = dL_dpix[pix];
            dLg[s] = dL_dpix[plane + pix];
            dLb[s] = dL_dpix[2 * plane + pix];
            if (DUAL) {
                dL2r[s] = dL_dpix2[pix];
                dL2g[s] = dL_dpix2[plane + pix];
                dL2b[s] = dL_dpix2[2 * plane + pix];
            }
        }
        pfx[s] = (float)px + off.x;
        pfy[s] = (float)py + off.y;
        tfb[s] = -T[s] * (bg0 * dLr[s] + bg1 * dLg[s] + bg2 * dLb[s]);  // -T_final * <bg, dL_dpixel>
        if (DUAL) { tfb2[s] = -T[s] * (bg0 * dL2r[s] + bg1 * dL2g[s] + bg2 * dL2b[s]); recd2[s] = 0.f; }   // both images sit on the same background
        recd[s] = 0.f;
        const float inf = __builtin_huge_valf();
        sb.x0[s] = wave_min_uniform(inside ? pfx[s] : inf);
        sb.x1[s] = wave_max_uniform(inside ? pfx[s] : -inf);
        sb.y0[s] = wave_min_uniform(inside ? pfy[s] : inf);
        sb.y1[s] = wave_max_uniform(inside ? pfy[s] : -inf);
    }

    // the last list position any pixel of a strip blended (wave-uniform: the largest n_contrib of the strip)
    int strip_last[4];
#pragma unroll
    for (int s = 0; s < 4; s++) {
        int m = last[s];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) m = max(m, __shfl_xor(m, d));
        strip_last[s] = __builtin_amdgcn_readfirstlane(m);
    }
    const uint2 range = ranges[tile];
    // the ten partial sums live across instances and are cleared after each reduction only: an instance without any
    // contributing lane (19 % of them) leaves them at zero
    // the ten sums live in five register PAIRS, so that clearing them after a reduction is five v_mov_b64 instead of ten v_mov_b32
    // (backward 0.4272 -> 0.4234 ms; the names below are the pairs' halves)
    f2 p0 = {0.f, 0.f}, p1 = {0.f, 0.f}, p2 = {0.f, 0.f}, p3 = {0.f, 0.f}, p4 = {0.f, 0.f};
    float ac2r = 0.f, ac2g = 0.f, ac2b = 0.f;   // DUAL: sum(w dL2_c)
#define acr p0.x
#define acg p0.y
#define acb p1.x
#define sx p1.y
#define sy p2.x
#define sab p2.y
#define sxx p3.x
#define sxy p3.y
#define syy p4.x
#define sq p4.y

#if WG_COUNT_PAIRS
    unsigned long long wgc[14] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    // LDSRED: this lane's read base (row 32 half + h, column m of its half; lanes with m >= 5 repeat columns 0 - 2), kept opaque so that it
    // stays in a register (rematerialised inside the loop it is three vector instructions per reduced instance)
    int rd0 = NS * (32 * red_half + (lane & 3)) + (red_m < NS ? red_m : red_m - NS);
    if (LDSRED) asm volatile("" : "+v"(rd0));
    for (int hi = hi0; hi > 0; hi -= BATCH) {
        // lane l stages the instance at list position hi-1-l (back to front, backward.cu:517)
        const int posl = hi - 1 - lane;
        __syncthreads();
        uint32_t mymask = 0;
        if (posl >= 0) {
            const uint32_t id = point_list[range.x + posl];
            float4 q0 = splats[3 * (size_t)id];
            float4 q1 = splats[3 * (size_t)id + 1];
            float4 q2 = splats[3 * (size_t)id + 2];
            mymask = strip_mask(q0, q1, q2, sb);
            const float3 c2 = make_float3(q1.z, q2.z, q2.w);   // DUAL: the second colour set, in the record's spare floats
            scale_conic(q0, q1);
            q2.z = 2.0f * q0.z;  // 2 ca, 2 cc: the gradient of the exponent, up to the factor 1 / log2(e) applied after the reduction
            q2.w = 2.0f * q1.x;
            // the record's spare float carries the Gaussian id to the reduction (no second LDS array, no LDS round trip on the reduction
            // path) -- on the default path as the BYTE OFFSET of its gradient record, so that the ten reducing lanes form their address with
            // one 32-bit add on top of the uniform base (48 P < 2^32) instead of a 64-bit multiply-add per reduced instance
            q1.z = __uint_as_float((RECORD && !DUAL && !DET) ? id * (uint32_t)(GRAD_REC_FLOATS * sizeof(float)) : id);
            if (DET) {  // deterministic mode: it carries the instance's SLOT instead -- the three gathers it takes are issued here, with
                        // the record's, by the staging lane, not behind the butterfly by the ten storing lanes (0.536 -> see EXPERIMENTS.md)
                const ushort4 rc = rects[id];
                q1.z = __uint_as_float(offsets_incl[id] - tiles_touched[id] + (uint32_t)((ty - rc.y) * (rc.z - rc.x) + (tx - rc.x)));
            }
            if (DUAL) lds[RS * lane + 3] = make_float4(c2.x, c2.y, c2.z, 0.f);
            lds[RS * lane] = q0;
            lds[RS * lane + 1] = q1;
            lds[RS * lane + 2] = q2;
        }
        __syncthreads();
        // the batch's strip masks as wave-uniform 64-bit words (see render_fwd.hip).  A strip none of whose pixels was still
        // accumulating at a list position cannot receive anything from it (backward.cu:536 skips per pixel): instance j of the
        // batch sits at position hi - 1 - j, so only j >= hi - strip_last[s] are of interest to strip s.
        uint64_t reach[4];
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const int first = hi - strip_last[s];  // wave-uniform
            const uint64_t live = first <= 0 ? ~0ull : (first >= 64 ? 0ull : (~0ull << first));
            reach[s] = __ballot((mymask >> s) & 1u) & live;
        }
        uint64_t todo = reach[0] | reach[1] | reach[2] | reach[3];
        while (todo != 0ull) {
            const int j = __builtin_ctzll(todo);
            todo &= todo - 1;
            const float4 r1 = lds[RS * j + 1];
            const int pos = hi - 1 - j;  // "contributor" after the decrement at backward.cu:531
            const float4 r0 = lds[RS * j];
            const SplatCoef sc = coef_of(r0, r1);
            const float o = sc.o;
            const float4 gb = lds[RS * j + 2];  // green, blue, 2 ca, 2 cc
            const float colr = r1.w, colg = gb.x, colb = gb.y;
            float4 col2 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (DUAL) col2 = lds[RS * j + 3];
            // Per-lane partial sums over this lane's (up to four) pixels.  Constant factors of the reference's
            // expressions are applied once, after the wave reduction:
            //   q = G * dL_dalpha;   u = A dx + B dy;   v = C dy + B dx        (dG/ddelx = -G u, dG/ddely = -G v)
            //   dL_dmean2D.x = -o * 0.5W * sum(q u)           dL_dconic.xx = -0.5 o * sum(q dx dx)
            //   dL_dmean2D.y = -o * 0.5H * sum(q v)           dL_dconic.xy = -0.5 o * sum(q dx dy)
            //   dL_dmean2D.z =  o * sum(|q| (0.5W |u| + 0.5H |v|))   dL_dconic.yy = -0.5 o * sum(q dy dy)
            //   dL_dopacity  = sum(q)
            // Predicates as wave-wide masks in scalar registers, built from the compares' own outputs (as in render_fwd.hip, round 5): the
            // contribution block is entered through one scalar AND on the execution mask, and "did anybody contribute" is a scalar compare.
            uint64_t any_m = 0ull;
            WG_CNT(0, 1);
#pragma unroll
            for (int s = 0; s < 4; s++) {
                if (((reach[s] >> j) & 1ull) == 0ull) continue;  // wave-uniform
                WG_CNT(1, 1);
                const uint64_t before_last_m = __builtin_amdgcn_ballot_w64(pos < last[s]);
                WG_CNT(2, __popcll(before_last_m));
                PairEval e;
                uint64_t pass_m;
                if (EXACT) {
                    float margin, band;
                    eval_alpha_banded(sc, pfx[s], pfy[s], e, margin, band);
                    pass_m = __builtin_amdgcn_ballot_w64(margin >= 0.0f);
                    const uint64_t fragile_m = before_last_m & __builtin_amdgcn_ballot_w64(fabsf(margin) < band);
                    if (fragile_m != 0ull) {  // rare: the reference's arithmetic on the record as preprocess wrote it
                        const size_t r = 3 * (size_t)point_list[range.x + pos];
                        float4 q0 = splats[r], q1 = splats[r + 1];
                        halve_conic(q0, q1);
                        float dx, dy, G, alpha;
                        const float power = eval_alpha_exact_values(exact_coef_of(q0, q1), pfx[s], pfy[s], dx, dy, G, alpha);
                        const uint64_t px_m = __builtin_amdgcn_ballot_w64(!(power > 0.0f)) & __builtin_amdgcn_ballot_w64(!(alpha < (1.0f / 255.0f)));
                        pass_m = (pass_m & ~fragile_m) | (px_m & fragile_m);
                        if (__builtin_amdgcn_inverse_ballot_w64(fragile_m)) { e.G = G; e.alpha = alpha; }
                    }
                } else {
                    const float p2 = eval_alpha_values(sc, pfx[s], pfy[s], e);
                    pass_m = __builtin_amdgcn_ballot_w64(!(p2 > 0.0f)) & __builtin_amdgcn_ballot_w64(!(e.alpha < (1.0f / 255.0f)));
                }
                const uint64_t go_m = before_last_m & pass_m;
                WG_CNT(3, __popcll(go_m));
                WG_CNT(5, go_m == 0ull ? 1 : 0);
#if WG_COUNT_PAIRS
                { const int c = __popcll(go_m); if (c) WG_CNT(c <= 8 ? 10 : c <= 24 ? 11 : c <= 48 ? 12 : 13, 1); }
#endif
                any_m |= go_m;
                if (__builtin_amdgcn_inverse_ballot_w64(go_m)) {
                    const float a = e.alpha;
                    const float inv = __builtin_amdgcn_rcpf(1.0f - a);
                    const float Tn = T[s] * inv;  // T / (1 - alpha), backward.cu:548
                    T[s] = Tn;
                    const float w = a * Tn;  // dchannel_dcolor
                    acr += w * dLr[s];
                    acg += w * dLg[s];
                    acb += w * dLb[s];
                    // sum_ch (c_ch - accum_rec_ch) * dL_ch, with accum_rec carried as its dot product with dL_dpixel:
                    // accum_rec' = alpha*c + (1-alpha)*accum_rec  (backward.cu:560)  =>  recd' = recd + alpha*(cd - recd)
                    const float cd = colr * dLr[s] + colg * dLg[s] + colb * dLb[s];
                    const float diff = cd - recd[s];
                    recd[s] += a * diff;
                    const float dLda = diff * Tn + tfb[s] * inv;
                    float q = e.G * dLda, qabs = fabsf(q);
                    if (DUAL) {
                        // the second set's own dL_dalpha: every sum below is linear in it EXCEPT the abs-gradient (backward.cu:593-595),
                        // which two separate calls accumulate as |g1| + |g2|, not |g1 + g2|
                        ac2r += w * dL2r[s];
                        ac2g += w * dL2g[s];
                        ac2b += w * dL2b[s];
                        const float cd2 = col2.x * dL2r[s] + col2.y * dL2g[s] + col2.z * dL2b[s];
                        const float diff2 = cd2 - recd2[s];
                        recd2[s] += a * diff2;
                        const float q2 = e.G * (diff2 * Tn + tfb2[s] * inv);
                        q += q2;
                        qabs += fabsf(q2);
                    }
                    // -log2(e) * (A dx + B dy) and -log2(e) * (C dy + B dx): the constant goes into vscale
                    const float u = gb.z * e.dx + r0.w * e.dy;
                    const float v = gb.w * e.dy + r0.w * e.dx;
                    sq += q;
                    sx += q * u;
                    sy += q * v;
                    sab += qabs * (ddelx_dx * fabsf(u) + ddely_dy * fabsf(v));
                    sxx += q * e.xx;
                    sxy += q * e.xy;
                    syy += q * e.yy;
                }
            }
            if (any_m == 0ull) continue;
            WG_CNT(4, 1);
#if WG_COUNT_PAIRS
            { const int c = __popcll(any_m); WG_CNT(c <= 1 ? 6 : c <= 4 ? 7 : c <= 16 ? 8 : 9, 1); }
#endif
            float total;
            if (LDSRED) {
                float* row = red + NS * lane;
                row[0] = pair_x32(acr, acg); row[1] = pair_x32(acb, sx); row[2] = pair_x32(sy, sab); row[3] = pair_x32(sxx, sxy); row[4] = pair_x32(syy, sq);
                if (DUAL) { row[5] = pair_x32(ac2r, ac2g); row[6] = pair_x32(ac2b, ac2b); }
                asm volatile("" ::: "memory");   // (compiler order only: one wave's LDS operations execute in program order)
                float t[8];
#pragma unroll
                for (int i = 0; i < 8; i++) t[i] = red[rd0 + 4 * NS * i];
                asm volatile("" ::: "memory");
#pragma unroll
                for (int d = 1; d < 8; d <<= 1)
#pragma unroll
                    for (int i = 0; i < 8; i += 2 * d) t[i] += t[i + d];
                total = t[0];
                total += dpp_f<0xB1>(total);
                total += dpp_f<0x4E>(total);
            } else {
                total = DUAL ? butterfly13(acr, acg, acb, sx, sy, sab, sxx, sxy, syy, sq, ac2r, ac2g, ac2b, lane)
                             : butterfly10(acr, acg, acb, sx, sy, sab, sxx, sxy, syy, sq, lane);
            }
            if (DET) {
                if (issue) {
                    const uint32_t slot = __float_as_uint(r1.z);
                    det_slots[(size_t)slot * SLOT_FLOATS + vidx] = total;
                    if (vidx == 0) det_flags[slot] = 1;  // only flagged slots hold sums: the slot array itself is never cleared
                }
            } else if (RECORD) {
                if (DUAL) { if (issue) unsafeAtomicAdd(abase + (size_t)__float_as_uint(r1.z) * astride, total); }
                else if (issue) unsafeAtomicAdd(reinterpret_cast<float*>(reinterpret_cast<char*>(grad_rec) + (size_t)(__float_as_uint(r1.z) + 4u * (uint32_t)vidx)), total);
            } else {
                if (issue) unsafeAtomicAdd(abase + astride * __float_as_uint(r1.z), total * (oscale ? (vidx == 5 ? fabsf(o) : o) * vscale : vscale));  // 4*P < 2^32
            }
            // (hipcc scalarises "p = {0, 0}" into two v_mov_b32: spelled out)
            asm volatile("v_mov_b64 %0, 0\n\tv_mov_b64 %1, 0\n\tv_mov_b64 %2, 0\n\tv_mov_b64 %3, 0\n\tv_mov_b64 %4, 0"
                         : "=v"(p0), "=v"(p1), "=v"(p2), "=v"(p3), "=v"(p4));
            if (DUAL) ac2r = ac2g = ac2b = 0.f;
        }
    }
#if WG_COUNT_PAIRS
    if (lane == 0)
        for (int i = 0; i < 14; i++) atomicAdd(&g_bwd_counters[i], wgc[i]);
#endif
#if WG_PROBE
    if (lane == 0 && blockIdx.x < 65536) {
        g_bwd_probe[4 * blockIdx.x] = probe_t0;
        g_bwd_probe[4 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
        g_bwd_probe[4 * blockIdx.x + 2] = ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32) | (unsigned)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
        g_bwd_probe[4 * blockIdx.x + 3] = (unsigned long long)tile;
    }
#endif
}

#undef acr
#undef acg
#undef acb
#undef sx
#undef sy
#undef sab
#undef sxx
#undef sxy
#undef syy
#undef sq

// Per-Gaussian sum of the deterministic mode, in a FIXED order.  The 64 Gaussians of a wave own ONE contiguous range of slots
// [prefix(first) .. prefix(last) + tiles_touched(last)) -- slot = exclusive prefix of tiles_touched + the tile's index inside the
// Gaussian's rectangle.  Only about a quarter of the slots were written (the per-tile pass reduces an instance only where a pixel
// contributed; it flags those), so the wave first COMPACTS: it walks its range's flag bytes 64 at a time (coalesced), and the flagged slots'
// indices are appended, in slot order, to a 64-entry batch in LDS.  A full batch (and the last, partial one) is reduced with every lane
// busy: lane l loads batch entry l's ten floats, finds its owner among the wave's Gaussians (a 6-step search over their end offsets), a
// segmented inclusive scan over the lanes (segment = run of equal owners) adds each Gaussian's slots of the batch in a fixed tree, the last
// lane of each segment hands the total to its owner's lane through LDS, which adds it to its accumulator -- batch after batch, in slot
// order.  The shape of the trees depends only on tiles_touched and on which slots were flagged (i.e. on the frame), never on timing:
// bit-identical gradients run to run.  The slot array is never cleared, only the 1-byte flags are; the kernel WRITES the whole 48-byte
// record, zeros for a Gaussian nothing was added to.
// (Rounds 3-4: one thread per Gaussian scanning its own flags and gathering its flagged slots: 121 us at the headline scene, one lane in ~4
// doing anything.  Round 5, first form: a segmented scan over ALL slots 64 at a time: 107 us -- 76 LDS shuffles per 64 slots of which 15 hold
// anything.  This form scans a quarter as many lanes: profiles/r5/ab_deterministic_backward*.)
// NF2 = float2 per slot: 5 (ten sums) or 7 (the two-colour walk's thirteen, padded to fourteen: sums 10, 11 go to the record's two spare
// floats, sum 12 to grad_aux[g] = grad_rec[12 P + g])
template <int NF2>
__global__ void __launch_bounds__(256) det_reduce_kernel(int P, const uint32_t* __restrict__ offsets_incl, const uint32_t* __restrict__ tiles_touched,
                                                         const float* __restrict__ det_slots, const unsigned char* __restrict__ det_flags,
                                                         float* __restrict__ grad_rec, size_t slot_capacity) {
    constexpr int NV = 2 * NF2;
    __shared__ uint32_t s_ends[4][64];
    __shared__ uint32_t s_batch[4][64];
    __shared__ float s_out[4][64][NV + 1];   // (+1: an odd row stride keeps the owners' reads off one bank)
    __shared__ unsigned char s_touch[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = g < P;
    const uint32_t n = valid ? tiles_touched[g] : 0u;
    const uint32_t end = valid ? offsets_incl[g] : 0u;
    const uint32_t base = end - n;
    const int g0 = blockIdx.x * blockDim.x + wave * 64;
    if (g0 >= P) return;  // wave-uniform
    const int last_lane = min(63, P - 1 - g0);
    const uint32_t wlo = (uint32_t)__builtin_amdgcn_readlane((int)base, 0);
    const uint32_t whi = (uint32_t)__builtin_amdgcn_readlane((int)end, last_lane);
    uint32_t* ends = s_ends[wave];
    uint32_t* batch = s_batch[wave];
    ends[lane] = valid ? end : 0xffffffffu;   // nondecreasing over the lanes (inclusive prefix sums); lanes past P: beyond every slot
    float acc[NV];
#pragma unroll
    for (int v = 0; v < NV; v++) acc[v] = 0.f;

    auto reduce_batch = [&](uint32_t cnt) {   // cnt (wave-uniform, 1..64) slot indices in batch[], in slot order
        __builtin_amdgcn_wave_barrier();
        const bool act = (uint32_t)lane < cnt;
        const uint32_t k = act ? batch[lane] : 0u;
        float x[NV];
        {
            const float2* sl = reinterpret_cast<const float2*>(det_slots + (size_t)k * NV);  // 40- / 56-byte slots: 8-byte aligned
#pragma unroll
            for (int h = 0; h < NF2; h++) {
                const float2 t = act ? sl[h] : make_float2(0.f, 0.f);
                x[2 * h] = t.x;
                x[2 * h + 1] = t.y;
            }
        }
        // owner = the wave's Gaussian whose slot range holds k = the number of end offsets <= k
        int owner = 0;
#pragma unroll
        for (int st = 32; st >= 1; st >>= 1)
            if (ends[owner + st - 1] <= k) owner += st;
        if (!act) owner = 64 + lane;   // idle lanes: segments of their own
        // (the two shuffles stand alone: behind a short-circuit `lane == 0 ||` the compiler runs them with that lane masked off, and a
        //  ds_bpermute returns 0 for a source lane that is not executing -- lane 1 then saw "owner 0" to its left and opened a segment)
        const int owner_left = __shfl_up(owner, 1), owner_right = __shfl_down(owner, 1);
        int head = (lane == 0 || owner_left != owner) ? 1 : 0;
        const bool seg_last = act && (lane == 63 || owner_right != owner);
        // segmented inclusive scan over the 64 lanes (Hillis-Steele; a lane stops taking once a head lies in its window)
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int hup = __shfl_up(head, d);
            float up[NV];
#pragma unroll
            for (int v = 0; v < NV; v++) up[v] = __shfl_up(x[v], d);
            if (lane >= d && head == 0) {
#pragma unroll
                for (int v = 0; v < NV; v++) x[v] += up[v];
                head = hup;
            }
        }
        s_touch[wave][lane] = 0;
        __builtin_amdgcn_wave_barrier();
        if (seg_last) {
#pragma unroll
            for (int v = 0; v < NV; v++) s_out[wave][owner][v] = x[v];
            s_touch[wave][owner] = 1;
        }
        __builtin_amdgcn_wave_barrier();
        if (s_touch[wave][lane]) {
#pragma unroll
            for (int v = 0; v < NV; v++) acc[v] += s_out[wave][lane][v];
        }
        __builtin_amdgcn_wave_barrier();
    };

    uint32_t nb = 0;   // slots in the batch (wave-uniform)
    constexpr int G = 4;   // flag bytes of G x 64 slots are fetched together: one memory round trip per 256 slots
    for (uint32_t glo = wlo; glo < whi; glo += 64u * G) {  // wave-uniform trip count
        bool flagged[G];
#pragma unroll
        for (int u = 0; u < G; u++) {
            const uint32_t k = glo + 64u * u + (uint32_t)lane;
            flagged[u] = k < whi && (size_t)k < slot_capacity && det_flags[k] != 0;
        }
#pragma unroll
        for (int u = 0; u < G; u++) {
            const uint64_t m = __ballot(flagged[u]);
            const uint32_t c = (uint32_t)__popcll(m);
            if (c == 0u) continue;   // wave-uniform
            if (nb + c > 64u) { reduce_batch(nb); nb = 0u; }
            if (flagged[u]) batch[nb + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = glo + 64u * u + (uint32_t)lane;
            nb += c;
        }
    }
    if (nb != 0u) reduce_batch(nb);
    if (valid) {
        float4* out = reinterpret_cast<float4*>(grad_rec + (size_t)g * GRAD_REC_FLOATS);
        out[0] = make_float4(acc[0], acc[1], acc[2], acc[3]);
        out[1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
        if (NF2 == 5) out[2] = make_float4(acc[8], acc[9], 0.f, 0.f);
        else {
            out[2] = make_float4(acc[8], acc[9], acc[NF2 == 5 ? 0 : 10], acc[NF2 == 5 ? 0 : 11]);
            grad_rec[(size_t)P * GRAD_REC_FLOATS + g] = acc[NF2 == 5 ? 0 : 12];
        }
    }
}

hipError_t launch_render_backward(int W, int H, int gx, int gy, const ImageState& img, const BinningState& b,
                                  const GeometryState& g, const float* subpixel_offset, const float* background,
                                  const float* dL_dpix, float* dL_dmean2D, float* dL_dconic, float* dL_dopacity,
                                  float* dL_dcolor, bool record, bool exact, const float* dL_dpix2, float* det_slots, unsigned char* det_flags, size_t slot_capacity, int P,
                                  hipStream_t stream) {
    const int tiles = gx * gy;
    if (tiles <= 0) return hipSuccess;
#define WG_LAUNCH3(REC, DET, EX, DU)                                                                                                            \
    hipLaunchKernelGGL((render_backward_kernel<REC, DET, EX, DU>), dim3(tiles), dim3(64), 0, stream, W, H, gx, tiles, img.order_bwd, img.ranges,     \
                       b.point_list, g.splats, reinterpret_cast<const float2*>(subpixel_offset), background, img.final_T, img.n_contrib,    \
                       img.tile_last, dL_dpix, dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor, g.grad_rec, g.rects, g.point_offsets,         \
                       g.tiles_touched, det_slots, det_flags, dL_dpix2, g.grad_rec + (size_t)P * GRAD_REC_FLOATS)
#define WG_LAUNCH2(REC, DET, EX) WG_LAUNCH3(REC, DET, EX, false)
#define WG_LAUNCH(REC, DET) do { if (exact) WG_LAUNCH2(REC, DET, true); else WG_LAUNCH2(REC, DET, false); } while (0)
    if (dL_dpix2 && det_slots) {   // two colour sets, deterministic: fourteen-float slots
        if (exact) WG_LAUNCH3(true, true, true, true); else WG_LAUNCH3(true, true, false, true);
        hipLaunchKernelGGL(det_reduce_kernel<7>, dim3((P + 255) / 256), dim3(256), 0, stream, P, g.point_offsets, g.tiles_touched, det_slots, det_flags,
                           g.grad_rec, slot_capacity);
    } else if (dL_dpix2) {   // (api.hip has checked: gradient record on)
        if (exact) WG_LAUNCH3(true, false, true, true); else WG_LAUNCH3(true, false, false, true);
    } else if (det_slots) {
        WG_LAUNCH(true, true);
        hipLaunchKernelGGL(det_reduce_kernel<5>, dim3((P + 255) / 256), dim3(256), 0, stream, P, g.point_offsets, g.tiles_touched, det_slots, det_flags,
                           g.grad_rec, slot_capacity);
    } else if (record) WG_LAUNCH(true, false);
    else WG_LAUNCH(false, false);
#undef WG_LAUNCH
#undef WG_LAUNCH2
#undef WG_LAUNCH3
    return hipGetLastError();
}

}  // namespace wg

#if WG_PROBE
extern "C" int wg_probe_fetch_bwd(void* dst, size_t bytes) {
    hipError_t e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpyFromSymbol(dst, HIP_SYMBOL(wg::g_bwd_probe), bytes < sizeof(wg::g_bwd_probe) ? bytes : sizeof(wg::g_bwd_probe), 0, hipMemcpyDeviceToHost);
    return (int)e;
}
#endif
#if WG_COUNT_PAIRS
extern "C" int wg_debug_bwd_counters(unsigned long long* out16, int reset) {
    hipError_t e = hipDeviceSynchronize();
    if (e == hipSuccess && out16) e = hipMemcpyFromSymbol(out16, HIP_SYMBOL(wg::g_bwd_counters), 16 * sizeof(unsigned long long));
    if (e == hipSuccess && reset) {
        const unsigned long long z[16] = {};
        e = hipMemcpyToSymbol(HIP_SYMBOL(wg::g_bwd_counters), z, sizeof(z));
    }
    return e == hipSuccess ? 0 : -3;
}
#endif
