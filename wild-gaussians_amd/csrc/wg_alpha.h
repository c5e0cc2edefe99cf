// Shared per-(pixel, splat) falloff evaluation and per-(tile, splat) strip culling, used by BOTH render kernels
// so that the forward and the backward pass take identical skip decisions (forward.cu:353-366 == backward.cu:536-546).
#pragma once
#include <hip/hip_runtime.h>

namespace wg {

constexpr float WG_LOG2E = 1.4426950408889634f;

struct SplatCoef {
    float mx, my;      // pixel-space mean
    float ca, cb, cc;  // log2(e) * (-0.5*conic.x, -conic.y, -0.5*conic.z)
    float o;           // opacity * mip-filter coef
};

// record layout in HBM (preprocess.hip): r0 = (mx, my, conic.x, conic.y), r1 = (conic.z, opacity, -, red), r2 = (green, blue, -, -).
// The render kernels park it in LDS with the conic already scaled for the exp2 (one lane does that once per staged instance;
// every wave-instruction of the per-instance loop that is saved there is saved for each visited instance):
//   r0 = (mx, my, ca, cb), r1 = (cc, opacity, -, red), r2 = (green, blue, 2 ca, 2 cc)   [r2.zw: backward only]
__device__ __forceinline__ void scale_conic(float4& r0, float4& r1) {
    r0.z = -0.5f * WG_LOG2E * r0.z;
    r0.w = -WG_LOG2E * r0.w;
    r1.x = -0.5f * WG_LOG2E * r1.x;
}
__device__ __forceinline__ SplatCoef coef_of(const float4 r0, const float4 r1) {  // from a parked record
    SplatCoef c;
    c.mx = r0.x;
    c.my = r0.y;
    c.ca = r0.z;
    c.cb = r0.w;
    c.cc = r1.x;
    c.o = r1.y;
    return c;
}

// Returns true when the pair passes the reference's two skips: power <= 0 and alpha >= 1/255.
// G = exp(power) (unclamped), alpha = min(0.99, o*G).  The three second-order products of the offset are returned
// because the backward pass needs them again for the conic gradient.
struct PairEval {
    float dx, dy, xx, xy, yy, G, alpha;
};
__device__ __forceinline__ bool eval_alpha(const SplatCoef& c, float pfx, float pfy, PairEval& e) {
    e.dx = c.mx - pfx;
    e.dy = c.my - pfy;
    e.xx = e.dx * e.dx;
    e.xy = e.dx * e.dy;
    e.yy = e.dy * e.dy;
    const float p2 = c.ca * e.xx + c.cb * e.xy + c.cc * e.yy;  // log2(e) * power
    e.G = __builtin_amdgcn_exp2f(p2);
    e.alpha = fminf(0.99f, c.o * e.G);
    return !(p2 > 0.0f) && !(e.alpha < (1.0f / 255.0f));   // the reference's comparisons (forward.cu:360,365): a NaN passes both
}

// eval_alpha() handing p2 (= log2(e) * power) back instead of the decision
__device__ __forceinline__ float eval_alpha_values(const SplatCoef& c, float pfx, float pfy, PairEval& e) {
    e.dx = c.mx - pfx;
    e.dy = c.my - pfy;
    e.xx = e.dx * e.dx;
    e.xy = e.dx * e.dy;
    e.yy = e.dy * e.dy;
    const float p2 = c.ca * e.xx + c.cb * e.xy + c.cc * e.yy;
    e.G = __builtin_amdgcn_exp2f(p2);
    e.alpha = fminf(0.99f, c.o * e.G);
    return p2;
}

// ---- decision-exact compositing (Options::exact_compositing, default on) ------------------------------------------------------------
// The skips of forward.cu:356-372 / backward.cu:536-546 are threshold decisions on float32 values; a value that differs from the
// reference's in its last bits lands on the other side of a threshold once in ~10^8 pairs, and one flipped decision moves a pixel by
// up to ~1e-2 (the instance at which a pixel stops is not blended at all) and T, n_contrib and every later decision of that pixel
// with it.  So the values the decisions are taken on are computed HERE WITH THE REFERENCE'S OWN ARITHMETIC, operation for operation:
//   power = -0.5f * (con.x * d.x * d.x + con.z * d.y * d.y) - con.y * d.x * d.y      (forward.cu:359, unfused, in that order)
//   alpha = min(0.99f, con.w * exp(power))                                             (forward.cu:364)
//   test_T = T * (1 - alpha)                                                           (forward.cu:367)
// with `exp` = the float32 expansion llvm emits for it on gfx950 (AMDGPULegalizerInfo / SITargetLowering: lowerFExp; it is what the
// reference's own sources compile to with hipcc and -ffp-contract=off -- the checker build the tests hold this to, whose disassembly shows it):
//   ph = x * log2e;  e = rint(ph);  pl = fma(x, log2e, -ph);  pl = fma(x, 0x1.4ae0bep-26, pl);  r = ldexp(v_exp_f32((ph - e) + pl), (int)e)
// The parked record carries -0.5 * conic.x and -0.5 * conic.z: a multiplication by -0.5 is exact and commutes with every rounding
// in the expression above, so  fl(fl(hA dx) dx) + fl(fl(hC dy) dy)  is bit for bit  -0.5f * (fl(fl(A dx) dx) + fl(fl(C dy) dy)).
struct ExactCoef {
    float mx, my;
    float hA, B, hC;  // -0.5 conic.x, conic.y, -0.5 conic.z
    float o;
};
__device__ __forceinline__ void halve_conic(float4& r0, float4& r1) {
    r0.z = -0.5f * r0.z;
    r1.x = -0.5f * r1.x;
}
__device__ __forceinline__ ExactCoef exact_coef_of(const float4 r0, const float4 r1) {  // from a record parked by halve_conic()
    ExactCoef c;
    c.mx = r0.x; c.my = r0.y; c.hA = r0.z; c.B = r0.w; c.hC = r1.x; c.o = r1.y;
    return c;
}
// exp(x) for x <= 0 as the reference build computes it.  The expansion's two range selects are dropped: x > 88.7 cannot occur, and
// below -103 (where it returns 0) the clamp keeps ldexp's result a denormal: either way alpha < 1/255 and the pair is skipped.
// (The clamp is not optional: for |x| >~ 2^23 the split's low part pl is the rounding error of x * log2e, as large as ulp(ph) / 2, and
//  v_exp_f32 of it would overflow.)
// A NaN power is KEPT, as the reference keeps it: `power > 0` is false, exp(NaN) = NaN, min(0.99f, NaN) = 0.99f -- the pair is blended at
// alpha 0.99 (forward.cu:358-366; experiments/at_05a7d0c/nan_min_probe.hip shows v_min_f32 doing that on gfx950).  v_max_f32 would swallow the
// NaN (it returns the other operand), so the first fused multiply-add below takes the UNCLAMPED argument: for x >= -104 that is the clamped
// one, bit for bit; for x < -104 it makes pl hugely negative, v_exp_f32 underflows to 0 and the result is 0 instead of a denormal -- alpha
// < 1/255 and the pair skipped either way --; for a NaN it puts the NaN back.  No instruction is spent on it.  (No record the preprocess
// kernel writes can produce one -- a NaN conic comes with radius 0 and is parked with opacity 0, a NaN mean has no tile -- so this is the
// reference's per-pair semantics kept for its own sake; a NaN OPACITY is what does reach the walk: strip_mask_exact below.)
// tests/test_exp_expansion.py holds this function to the compiler's own `expf` (built -ffp-contract=off) over a dense range of arguments:
// a toolchain whose exp lowering changes fails that test directly, not only the parity tests downstream.
__device__ __forceinline__ float ref_expf_nonpos(float x) {
#pragma clang fp contract(off)
    const float xc = fmaxf(x, -104.0f);
    const float c = 0x1.715476p+0f, cc = 0x1.4ae0bep-26f;
    const float ph = xc * c;
    const float e = __builtin_rintf(ph);
    float pl = __builtin_fmaf(x, c, -ph);   // x, not xc: see above
    pl = __builtin_fmaf(xc, cc, pl);
    const float a = (ph - e) + pl;
    return __builtin_ldexpf(__builtin_amdgcn_exp2f(a), (int)e);
}
// the reference's power for a pixel at (pfx, pfy); d = mean - pixel (forward.cu:357)
__device__ __forceinline__ float ref_power(const ExactCoef& c, float dx, float dy) {
#pragma clang fp contract(off)
    const float t2 = (c.hA * dx) * dx;
    const float t4 = (c.hC * dy) * dy;
    const float s = t2 + t4;
    const float t6 = (c.B * dx) * dy;
    return s - t6;
}
// The same evaluation handing the two decided-upon VALUES back (-> power; alpha by reference): render_fwd.hip turns the compares into
// wave-wide masks itself.
__device__ __forceinline__ float eval_alpha_exact_values(const ExactCoef& c, float pfx, float pfy, float& dx, float& dy, float& G, float& alpha) {
#pragma clang fp contract(off)
    dx = c.mx - pfx;
    dy = c.my - pfy;
    const float power = ref_power(c, dx, dy);
    G = ref_expf_nonpos(power);
    alpha = fminf(0.99f, c.o * G);
    return power;
}
// Returns the reference's decision for the pair (power <= 0 and alpha >= 1/255) and its alpha, G = exp(power), dx, dy.
__device__ __forceinline__ bool eval_alpha_exact(const ExactCoef& c, float pfx, float pfy, float& dx, float& dy, float& G, float& alpha) {
#pragma clang fp contract(off)
    dx = c.mx - pfx;
    dy = c.my - pfy;
    const float power = ref_power(c, dx, dy);
    G = ref_expf_nonpos(power);
    alpha = fminf(0.99f, c.o * G);
    return !(power > 0.0f) && !(alpha < (1.0f / 255.0f));
}

// The fast evaluation (eval_alpha) with the distance of its two decisions from their thresholds:
//   margin = min(255 alpha - 1, -p2)   (>= 0: the pair passes both skips),    band = BAND_C |S| + BAND_C0,  S = ca xx + cc yy.
// |margin| >= band  =>  the reference's arithmetic takes the same decision, because the two evaluations cannot differ by more:
//   * exponent, in log2 units: the fast form rounds 5 times, the reference's 4 (+ the conversion), each by <= 2^-24 of the term it
//     forms, so |p2 - log2(e) power_ref| <= 9 * 2^-24 * M,  M = |ca| xx + |cb xy| + |cc| yy = |S| + |cb xy| <= 2 |S| + |p2|;
//     a pair near either threshold has |p2| <= log2(255) + 0.1, hence <= 1.1e-6 |S| + 4.4e-6;
//   * v_exp_f32 against the expansion's own v_exp_f32 + ldexp, o * G, 255 alpha - 1: 1 ulp each, <= 6e-7 in log2 units;
//   * 255 alpha - 1 moves by ln 2 (< 1) times the exponent's error.
// The constants carry a factor 2 on top.  (margin clearly negative on one side decides the pair whatever the other side says.)
constexpr float BAND_C = 2.5e-6f, BAND_C0 = 1.0e-5f;
__device__ __forceinline__ bool eval_alpha_banded(const SplatCoef& c, float pfx, float pfy, PairEval& e, float& margin, float& band) {
    e.dx = c.mx - pfx;
    e.dy = c.my - pfy;
    e.xx = e.dx * e.dx;
    e.xy = e.dx * e.dy;
    e.yy = e.dy * e.dy;
    const float S = __builtin_fmaf(c.ca, e.xx, c.cc * e.yy);
    const float p2 = __builtin_fmaf(c.cb, e.xy, S);
    e.G = __builtin_amdgcn_exp2f(p2);
    e.alpha = fminf(0.99f, c.o * e.G);
    band = __builtin_fmaf(BAND_C, fabsf(S), BAND_C0);
    margin = fminf(__builtin_fmaf(e.alpha, 255.0f, -1.0f), -p2);
    return margin >= 0.0f;
}

__device__ __forceinline__ float ref_test_T(float T, float alpha) {
#pragma clang fp contract(off)
    const float om = 1.0f - alpha;
    return T * om;
}

// XCD-aware block -> tile map: blocks are dealt round-robin to the 8 XCDs (block b runs on XCD b % 8), so
// XCD x receives the contiguous band of tiles [x*q + min(x,rem), ...): its private L2 only ever sees the
// splat records of that band.  Bijective for any tile count.
__device__ __forceinline__ int xcd_tile(int b, int tiles) {
    const int q = tiles >> 3, rem = tiles & 7;
    const int xcd = b & 7, i = b >> 3;
    return xcd * q + min(xcd, rem) + i;
}

// ---- wave64 min / max with DPP row operations; result valid in lane 63, returned as a wave-uniform value ----
#define WG_DPP_STEP(op, ctrl, rmask)                                                                                   \
    v = op(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), \
                                                                    ctrl, rmask, 0xF, false)))
__device__ __forceinline__ float wave_min_uniform(float v) {
    WG_DPP_STEP(fminf, 0xB1, 0xF); WG_DPP_STEP(fminf, 0x4E, 0xF); WG_DPP_STEP(fminf, 0x124, 0xF); WG_DPP_STEP(fminf, 0x128, 0xF);
    WG_DPP_STEP(fminf, 0x142, 0xA); WG_DPP_STEP(fminf, 0x143, 0xC);
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_max_uniform(float v) {
    WG_DPP_STEP(fmaxf, 0xB1, 0xF); WG_DPP_STEP(fmaxf, 0x4E, 0xF); WG_DPP_STEP(fmaxf, 0x124, 0xF); WG_DPP_STEP(fmaxf, 0x128, 0xF);
    WG_DPP_STEP(fmaxf, 0x142, 0xA); WG_DPP_STEP(fmaxf, 0x143, 0xC);
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
#undef WG_DPP_STEP

// Lane -> pixel map of the two render kernels: a lane owns one pixel in each of the tile's four "strips" (the unit of the
// per-instance culling below: an instance is evaluated on a strip only when its alpha >= 1/255 ellipse can reach it).
// WG_STRIP_QUADS = 1: the strips are the tile's four 8x8 quadrants (lane = x & 7, y & 7); 0: four 16x4 row bands (lane = column,
// row & 3).  Squarer strips are reached less often by a compact footprint: 2.75 instead of 2.95 strips per reached instance at the
// bench scene (a Monte-Carlo count over the scene's projected ellipses), at 32-byte instead of 64-byte row segments of pixel I/O.
#ifndef WG_STRIP_QUADS
#define WG_STRIP_QUADS 1
#endif
__device__ __forceinline__ int strip_x(int lane, int s) { return WG_STRIP_QUADS ? 8 * (s & 1) + (lane & 7) : (lane & 15); }
__device__ __forceinline__ int strip_y(int lane, int s) { return WG_STRIP_QUADS ? 8 * (s >> 1) + (lane >> 3) : (lane >> 4) + 4 * s; }

// Sample-position bounding boxes of the four strips of a tile (wave-uniform).
struct StripBounds {
    float x0[4], x1[4], y0[4], y1[4];
};

// One wave's compositing state for a tile (render_fwd.hip): 4 pixels per lane.
struct FwdTile {
    float pfx[4], pfy[4], T[4], Cr[4], Cg[4], Cb[4];
    float C2r[4], C2g[4], C2b[4];   // second colour set (DUAL walks only; never touched otherwise)
    uint32_t last[4];
    uint32_t alive;         // bit s: this lane's pixel of strip s is still accumulating
    uint32_t strips_alive;  // wave-uniform: strips with at least one such pixel
    StripBounds sb;
    int x0, y0;  // the tile's origin
};

// Conservative per-(splat, strip) reachability.  A pixel can only pass alpha >= 1/255 when power >= -ln(255*o), i.e. inside the
// ellipse d^T Q d <= 2 ln(255 o); splat_extent() is that ellipse's axis-aligned half-extent (inflated by 0.1% + 0.01 px so float
// rounding can never exclude a pixel the exact test would keep), computed once per Gaussian by the preprocess kernel and kept in
// the record's two spare floats (r2.z, r2.w); strip_mask() tests the box against each strip's sample-position box while staging.
// A splat below the 1/255 cut everywhere gets -inf (no strip), one with an indefinite conic +inf (all strips).
__device__ __forceinline__ float2 splat_extent(float A, float B, float C, float o) {
    if (!(o * 1.001f >= (1.0f / 255.0f))) return make_float2(-__builtin_huge_valf(), -__builtin_huge_valf());
    const float det = A * C - B * B;
    if (!(det > 0.0f) || !(A > 0.0f) || !(C > 0.0f)) return make_float2(__builtin_huge_valf(), __builtin_huge_valf());
    const float tau2 = 2.0f * 0.6931471805599453f * __builtin_amdgcn_logf(255.0f * o) * 1.002f + 0.002f;  // 2 ln(255 o), inflated
    const float idet = 1.0f / det;
    return make_float2(__builtin_sqrtf(fmaxf(tau2 * C * idet, 0.0f)) + 0.01f, __builtin_sqrtf(fmaxf(tau2 * A * idet, 0.0f)) + 0.01f);
}
__device__ __forceinline__ uint32_t strip_mask_box(const float4 r0, const float4 r2, const StripBounds& sb) {
    const float mx = r0.x, my = r0.y, ex = r2.z, ey = r2.w;
    uint32_t m = 0;
#pragma unroll
    for (int s = 0; s < 4; s++)
        if (mx + ex >= sb.x0[s] && mx - ex <= sb.x1[s] && my + ey >= sb.y0[s] && my - ey <= sb.y1[s]) m |= 1u << s;
    return m;
}

// The same question answered exactly: does the ellipse itself (not its box) meet the strip's sample box?  f(d) = A dx^2 + 2 B dx dy +
// C dy^2 is convex with its minimum at the mean, so its minimum over a box is attained on the box sides that face the mean; with
// xf = clamp(0, X0, X1) (the box's x nearest the mean; 0 when the box straddles it) and yf likewise, the two lines x = xf and
// y = yf inside the box contain those sides, and a 1-D clamped minimisation along each gives the minimum (0 when the mean is inside).
// One lane does this once per staged instance (~1.5 wave-instructions per instance), and the walk evaluates 16 % fewer strips and
// enters 11 % fewer instances than with the box test at the bench scene (tests/tools/strip_reach_estimate.py: 2.13 -> 1.78 strips per
// visited instance; 1.78 is also the count of strips with a passing pixel, i.e. nothing is left to cull at this granularity).
// Conservative by construction: tau^2 carries the same inflation as splat_extent(), plus a rounding allowance of 2e-6 of the largest
// magnitude the three products can take anywhere in the tile (both this evaluation and the per-pixel one round each product to
// ~1e-7 of it); a conic that is not safely positive definite reaches every strip.
#ifndef WG_STRIP_EXACT
#define WG_STRIP_EXACT 1
#endif
__device__ __forceinline__ uint32_t strip_mask_exact(const float4 r0, const float4 r1, const StripBounds& sb) {
    const float mx = r0.x, my = r0.y, A = r0.z, B = r0.w, C = r1.x, o = r1.y;
    const bool vis = !(o * 1.001f < (1.0f / 255.0f));  // true for a NaN opacity: the reference blends it at alpha = min(0.99f, NaN) = 0.99f everywhere (forward.cu:364)
    const float AC = A * C;
    const bool definite = A > 0.0f && C > 0.0f && (AC - B * B) > 4e-6f * AC;
    // A NaN opacity reaches every strip: the reference blends such a Gaussian at alpha = min(0.99f, NaN) = 0.99f on every pixel of its tile
    // rectangle (forward.cu:364).  (A NaN CONIC never gets here with an opacity: its radius is (int)NaN = 0, the reference's duplicateWithKeys
    // emits nothing for it -- rasterizer_impl.cu:85 -- and preprocess.hip parks such a record with opacity 0.)
    const bool o_nan = !(o == o);
    const float tau2 = 2.0f * 0.6931471805599453f * __builtin_amdgcn_logf(255.0f * o) * 1.002f + 0.002f;
    const float nBC = -B * __builtin_amdgcn_rcpf(C), nBA = -B * __builtin_amdgcn_rcpf(A), B2 = 2.0f * B;
    // the tile's sample box (wave-uniform; +-inf bounds of strips outside the image drop out of the min / max)
    const float bx0 = fminf(fminf(sb.x0[0], sb.x0[1]), fminf(sb.x0[2], sb.x0[3])), bx1 = fmaxf(fmaxf(sb.x1[0], sb.x1[1]), fmaxf(sb.x1[2], sb.x1[3]));
    const float by0 = fminf(fminf(sb.y0[0], sb.y0[1]), fminf(sb.y0[2], sb.y0[3])), by1 = fmaxf(fmaxf(sb.y1[0], sb.y1[1]), fmaxf(sb.y1[2], sb.y1[3]));
    const float Mx = fmaxf(fabsf(bx0 - mx), fabsf(bx1 - mx)), My = fmaxf(fabsf(by0 - my), fabsf(by1 - my));
    const float thr = tau2 + 2e-6f * (A * Mx * Mx + fabsf(B2) * Mx * My + C * My * My);
    uint32_t m = 0;
#pragma unroll
    for (int s = 0; s < 4; s++) {
        if (!(sb.x0[s] <= sb.x1[s])) continue;  // wave-uniform: no pixel of the strip is inside the image
        const float X0 = sb.x0[s] - mx, X1 = sb.x1[s] - mx, Y0 = sb.y0[s] - my, Y1 = sb.y1[s] - my;
        const float xf = __builtin_amdgcn_fmed3f(X0, 0.0f, X1), yf = __builtin_amdgcn_fmed3f(Y0, 0.0f, Y1);
        const float t = __builtin_amdgcn_fmed3f(nBC * xf, Y0, Y1);  // argmin of f(xf, .) over the box
        const float fv = xf * (A * xf + B2 * t) + C * t * t;
        const float u = __builtin_amdgcn_fmed3f(nBA * yf, X0, X1);  // argmin of f(., yf)
        const float fh = yf * (C * yf + B2 * u) + A * u * u;
        if (fminf(fv, fh) <= thr) m |= 1u << s;
    }
    return o_nan ? 15u : (vis ? (definite ? m : 15u) : 0u);
}
// r0, r1, r2: the record as preprocess wrote it (conic unscaled)
__device__ __forceinline__ uint32_t strip_mask(const float4 r0, const float4 r1, const float4 r2, const StripBounds& sb) {
    return WG_STRIP_EXACT ? strip_mask_exact(r0, r1, sb) : strip_mask_box(r0, r2, sb);
}

}  // namespace wg
