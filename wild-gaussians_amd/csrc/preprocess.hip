// K1 (per-Gaussian preprocess) and K12 (markVisible) for gfx950.
//
// Replaces preprocessCUDA / computeCov3D / computeCov2D / computeColorFromSH (forward.cu:20-268),
// in_frustum / getRect / ndc2Pix (auxiliary.h:41-56,139-164) and checkFrustum (rasterizer_impl.cu:54-66).
//
// This translation unit is compiled with -ffp-contract=off and spells every float operation in the
// operation order the reference source implies (including its silent double promotions), so that the
// integer results that drive everything downstream -- radii, tile rectangles, tiles_touched, depth
// keys -- are bit-identical to the CPU oracle's.  It is a streaming kernel: ~60-250 B read and
// ~70 B written per Gaussian, no reuse, so the work is laid out one Gaussian per lane with the
// per-Gaussian outputs packed into one 48-byte record that the render kernels gather in one go.
#include "wg_common.h"
#include "wg_alpha.h"
#include "wg_act.h"
#include "wg_sort.h"   // depth_code (sh_colour_kernel: near or far?)

#pragma clang fp contract(off)

namespace wg {

__constant__ const float SH_C0 = 0.28209479177387814f;
__constant__ const float SH_C1 = 0.4886025119029199f;
__constant__ const float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                     -1.0925484305920792f, 0.5462742152960396f};
__constant__ const float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                                     -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f};

// SH -> RGB for one channel (forward.cu:30-63); sh points at coefficient 0 of this channel, stride 3.
__device__ __forceinline__ float sh_channel(int deg, const float* sh, float x, float y, float z) {
    float result = SH_C0 * sh[0];
    if (deg > 0) {
        result = result - SH_C1 * y * sh[3] + SH_C1 * z * sh[6] - SH_C1 * x * sh[9];
        if (deg > 1) {
            float xx = x * x, yy = y * y, zz = z * z;
            float xy = x * y, yz = y * z, xz = x * z;
            result = result + SH_C2[0] * xy * sh[12] + SH_C2[1] * yz * sh[15] + SH_C2[2] * (2.0f * zz - xx - yy) * sh[18] +
                     SH_C2[3] * xz * sh[21] + SH_C2[4] * (xx - yy) * sh[24];
            if (deg > 2) {
                result = result + SH_C3[0] * y * (3.0f * xx - yy) * sh[27] + SH_C3[1] * xy * z * sh[30] +
                         SH_C3[2] * y * (4.0f * zz - xx - yy) * sh[33] + SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * sh[36] +
                         SH_C3[4] * x * (4.0f * zz - xx - yy) * sh[39] + SH_C3[5] * z * (xx - yy) * sh[42] +
                         SH_C3[6] * x * (xx - 3.0f * yy) * sh[45];
            }
        }
    }
    return result + 0.5f;
}

// FAST_SH (M == 16, 16-byte aligned): the wave moves its 64 Gaussians' 12 KiB SH block with coalesced 16-byte loads
// and transposes it through LDS (pitch 13 float4, conflict-free b128 accesses) instead of 48 loads at a 192-byte
// lane stride.  The values, and the order of the arithmetic on them, are unchanged.
constexpr int SH_PITCH4 = 13;

// GEOM_ONLY (round 6, frames that attempt the near / far split with plain SH colours): everything but the colour -- the record's colour floats
// are left zero, the clamp flags cleared and the SH block is not read; sh_colour_kernel below colours the Gaussians whose instances can be
// walked at all: the NEAR ones before the binning chain, the far ones only when a tile asks for its far instances.  The 192-byte SH block is
// four fifths of what this kernel reads per Gaussian, and at 10 M Gaussians / 4K nine Gaussians in ten are far.
template <int SH_MODE, bool PRECOMP, bool TONE, bool GEOM_ONLY = false>   // SH_MODE: 0 generic layout, 1 coalesced block through LDS, 2 the same with non-temporal loads
__global__ void __launch_bounds__(64) preprocess_kernel(FwdParams p, GeometryState g, int* __restrict__ radii_out, ToneArg<TONE> tone) {
    static_assert(!GEOM_ONLY || (SH_MODE == 0 && !TONE), "the geometry-only instantiation loads no SH block");
    constexpr bool FAST_SH = SH_MODE != 0, NT = SH_MODE == 2;
    __shared__ float4 stage[FAST_SH ? 64 * SH_PITCH4 : 1];
    const int lane = threadIdx.x;
    const int base = blockIdx.x * 64;
    const int idx = base + lane;
    const bool inside = idx < p.P;
    const int ld = inside ? idx : p.P - 1;  // out-of-range lanes of the last wave shadow the last Gaussian and store nothing

    // All of this wave's loads are issued up front: the 12 KiB SH block stays in flight in registers while the geometry
    // below is computed, and only then goes through LDS.
    // camera constants into scalar registers before the ordering fences below (which would turn them into vector loads)
    float vm[16], pm[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
        vm[i] = p.viewmatrix[i];
        pm[i] = p.projmatrix[i];
    }
    // the camera position only enters through the SH view direction (forward.cu:33): absent (null) with precomputed colours
    float camx = 0.f, camy = 0.f, camz = 0.f;
    if (!GEOM_ONLY && p.colors_precomp == nullptr) { camx = p.cam_pos[0]; camy = p.cam_pos[1]; camz = p.cam_pos[2]; }  // wave-uniform, scalar loads
    // Geometry inputs first, SH block second: the memory counter retires loads in issue order, so whatever the geometry
    // waits for has to be issued ahead of the twelve SH loads for those to stay in flight behind it.
    float px = p.means3D[3 * ld], py = p.means3D[3 * ld + 1], pz = p.means3D[3 * ld + 2];
    float opacity = p.opacities[ld];
    // raw-parameter mode: the 3-D filter's value, loaded with the rest (through a pointer that is always valid: a branch around a
    // load here would be a join the compiler waits at for every load in flight)
    float filt = (p.filter_3D ? p.filter_3D : p.opacities)[ld];
    float sc0 = 0.f, sc1 = 0.f, sc2 = 0.f;
    float4 quat = make_float4(0.f, 0.f, 0.f, 0.f);
    float c0 = 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f, c4 = 0.f, c5 = 0.f;
    if (!PRECOMP) {  // a template parameter, not a branch: a join here makes the compiler wait for every load in flight
        sc0 = p.scales[3 * ld]; sc1 = p.scales[3 * ld + 1]; sc2 = p.scales[3 * ld + 2];
        quat = reinterpret_cast<const float4*>(p.rotations)[ld];
    } else {
        const float* c = p.cov3D_precomp + 6 * (size_t)ld;
        c0 = c[0]; c1 = c[1]; c2 = c[2]; c3 = c[3]; c4 = c[4]; c5 = c[5];
    }
    // (named registers, not an array: hipcc leaves a 12 x float4 array in scratch)
    float4 sr0, sr1, sr2, sr3, sr4, sr5, sr6, sr7, sr8, sr9, sr10, sr11;
    sr0 = sr1 = sr2 = sr3 = sr4 = sr5 = sr6 = sr7 = sr8 = sr9 = sr10 = sr11 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (FAST_SH) {
        const float4* src = reinterpret_cast<const float4*>(p.shs) + (size_t)base * 12;
        const int last = min(64, p.P - base) * 12 - 1;
#define WG_SH_LOAD(i) sr##i = stream_load4<NT>(&src[min(i * 64 + lane, last)]);
        WG_SH_LOAD(0) WG_SH_LOAD(1) WG_SH_LOAD(2) WG_SH_LOAD(3) WG_SH_LOAD(4) WG_SH_LOAD(5)
        WG_SH_LOAD(6) WG_SH_LOAD(7) WG_SH_LOAD(8) WG_SH_LOAD(9) WG_SH_LOAD(10) WG_SH_LOAD(11)
#undef WG_SH_LOAD
    }
    // pin the order: nothing of the geometry below may be scheduled ahead of the SH loads, nor the LDS staging ahead of it
    asm volatile("" : "+v"(px), "+v"(py), "+v"(pz), "+v"(filt) : : "memory");
    if (!PRECOMP && p.filter_3D != nullptr) {   // wave-uniform: get_gaussians() (method.py:1060-1086) on the raw parameters
        const ActFwd a = act_forward(quat, sc0, sc1, sc2, opacity, filt);
        quat = a.q;
        sc0 = a.sc[0]; sc1 = a.sc[1]; sc2 = a.sc[2];
        opacity = a.o * a.coef;
    }

    // forward.cu:200-201
    int radius_i = 0;
    uint32_t touched = 0;
    bool vis = false;
    float pixx = 0.f, pixy = 0.f, conx = 0.f, cony = 0.f, conz = 0.f, coef = 0.f;
    int rminx = 0, rminy = 0, rmaxx = 0, rmaxy = 0;

    // in_frustum: auxiliary.h:152-163 (near cull only)
    const float vx = vm[0] * px + vm[4] * py + vm[8] * pz + vm[12];
    const float vy = vm[1] * px + vm[5] * py + vm[9] * pz + vm[13];
    const float vz = vm[2] * px + vm[6] * py + vm[10] * pz + vm[14];
    const bool alive = inside && !(vz <= 0.2f);
    if (inside && !alive && p.prefiltered) {
        // auxiliary.h:156-160: the reference printf()s and __trap()s; same contract here.
        printf("Point is filtered although prefiltered is set. This shouldn't happen!");
        __builtin_trap();
    }

    if (alive) {
        // forward.cu:209-212
        const float hx = pm[0] * px + pm[4] * py + pm[8] * pz + pm[12];
        const float hy = pm[1] * px + pm[5] * py + pm[9] * pz + pm[13];
        const float hw = pm[3] * px + pm[7] * py + pm[11] * pz + pm[15];
        const float p_w = 1.0f / (hw + 0.0000001f);
        const float projx = hx * p_w, projy = hy * p_w;

        // ---- 3D covariance (forward.cu:129-163) ----
        if (!PRECOMP) {
            const float s0 = p.scale_modifier * sc0, s1 = p.scale_modifier * sc1, s2 = p.scale_modifier * sc2;
            const float4 q = quat;
            const float r = q.x, x = q.y, y = q.z, z = q.w;
            // column-major R as filled by the reference (forward.cu:145-149); M[c][r] = s_r * R[c][r]
            const float M00 = s0 * (1.f - 2.f * (y * y + z * z)), M01 = s1 * (2.f * (x * y - r * z)), M02 = s2 * (2.f * (x * z + r * y));
            const float M10 = s0 * (2.f * (x * y + r * z)), M11 = s1 * (1.f - 2.f * (x * x + z * z)), M12 = s2 * (2.f * (y * z - r * x));
            const float M20 = s0 * (2.f * (x * z - r * y)), M21 = s1 * (2.f * (y * z + r * x)), M22 = s2 * (1.f - 2.f * (x * x + y * y));
            // Sigma[c][r] = M[r][0]*M[c][0] + M[r][1]*M[c][1] + M[r][2]*M[c][2]
            c0 = M00 * M00 + M01 * M01 + M02 * M02;
            c1 = M10 * M00 + M11 * M01 + M12 * M02;
            c2 = M20 * M00 + M21 * M01 + M22 * M02;
            c3 = M10 * M10 + M11 * M11 + M12 * M12;
            c4 = M20 * M10 + M21 * M11 + M22 * M12;
            c5 = M20 * M20 + M21 * M21 + M22 * M22;
            float* dst = g.cov3D + 6 * (size_t)idx;
            dst[0] = c0; dst[1] = c1; dst[2] = c2; dst[3] = c3; dst[4] = c4; dst[5] = c5;
        }

        // ---- EWA 2D covariance + mip filter (forward.cu:74-124) ----
        const float limx = 1.3f * p.tan_fovx, limy = 1.3f * p.tan_fovy;
        const float txtz = vx / vz, tytz = vy / vz;
        const float tx = fminf(limx, fmaxf(-limx, txtz)) * vz;
        const float ty = fminf(limy, fmaxf(-limy, tytz)) * vz;
        const float j00 = p.focal_x / vz, j02 = -(p.focal_x * tx) / (vz * vz);
        const float j11 = p.focal_y / vz, j12 = -(p.focal_y * ty) / (vz * vz);
        // T = W*J with W[c][r] = vm[c + 4r]
        const float T00 = vm[0] * j00 + vm[2] * j02, T01 = vm[4] * j00 + vm[6] * j02, T02 = vm[8] * j00 + vm[10] * j02;
        const float T10 = vm[1] * j11 + vm[2] * j12, T11 = vm[5] * j11 + vm[6] * j12, T12 = vm[9] * j11 + vm[10] * j12;
        // A[c][r] = T[r][0]*V[0][c] + T[r][1]*V[1][c] + T[r][2]*V[2][c]  (V symmetric from cov3D)
        const float A00 = T00 * c0 + T01 * c1 + T02 * c2, A10 = T00 * c1 + T01 * c3 + T02 * c4, A20 = T00 * c2 + T01 * c4 + T02 * c5;
        const float A01 = T10 * c0 + T11 * c1 + T12 * c2, A11 = T10 * c1 + T11 * c3 + T12 * c4, A21 = T10 * c2 + T11 * c4 + T12 * c5;
        float cov00 = A00 * T00 + A10 * T01 + A20 * T02;
        const float cov01 = A01 * T00 + A11 * T01 + A21 * T02;
        float cov11 = A01 * T10 + A11 * T11 + A21 * T12;

        // forward.cu:112-118 -- max(1e-6, float) and "+1e-6" are double arithmetic in the reference
        const float det_0 = (float)fmax(1e-6, (double)(cov00 * cov11 - cov01 * cov01));
        const float det_1 = (float)fmax(1e-6, (double)((cov00 + p.kernel_size) * (cov11 + p.kernel_size) - cov01 * cov01));
        coef = (float)sqrt((double)det_0 / ((double)det_1 + 1e-6) + 1e-6);
        if ((double)det_0 <= 1e-6 || (double)det_1 <= 1e-6) coef = 0.0f;
        cov00 += p.kernel_size;
        cov11 += p.kernel_size;

        // forward.cu:231-249
        const float det = cov00 * cov11 - cov01 * cov01;
        if (det != 0.0f) {
            const float det_inv = 1.f / det;
            conx = cov11 * det_inv; cony = -cov01 * det_inv; conz = cov00 * det_inv;
            const float mid = 0.5f * (cov00 + cov11);
            const float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
            const float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
            const float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
            // ndc2Pix, auxiliary.h:41-44 (double)
            pixx = (float)((((double)projx + 1.0) * (double)p.W - 1.0) * 0.5);
            pixy = (float)((((double)projy + 1.0) * (double)p.H - 1.0) * 0.5);
            // getRect, auxiliary.h:46-56
            const int mr = (int)my_radius;
            rminx = min(p.gx, max(0, (int)((pixx - mr) / TILE_X)));
            rminy = min(p.gy, max(0, (int)((pixy - mr) / TILE_Y)));
            rmaxx = min(p.gx, max(0, (int)((pixx + mr + TILE_X - 1) / TILE_X)));
            rmaxy = min(p.gy, max(0, (int)((pixy + mr + TILE_Y - 1) / TILE_Y)));
            const int ntiles = (rmaxx - rminx) * (rmaxy - rminy);
            if (ntiles != 0) {
                vis = true;
                radius_i = mr;
                touched = (uint32_t)ntiles;
            }
        }
    }

    asm volatile("" : "+v"(pixx), "+v"(radius_i) : : "memory");
    if (FAST_SH) {
        const int nvalid = min(64, p.P - base) * 12;
#define WG_SH_STAGE(i)                                                          \
    {                                                                           \
        const int f = i * 64 + lane;                                            \
        if (f < nvalid) stage[(f / 12) * SH_PITCH4 + (f % 12)] = sr##i;         \
    }
        WG_SH_STAGE(0) WG_SH_STAGE(1) WG_SH_STAGE(2) WG_SH_STAGE(3) WG_SH_STAGE(4) WG_SH_STAGE(5)
        WG_SH_STAGE(6) WG_SH_STAGE(7) WG_SH_STAGE(8) WG_SH_STAGE(9) WG_SH_STAGE(10) WG_SH_STAGE(11)
#undef WG_SH_STAGE
        __syncthreads();
    }
    if (!inside) return;

    if (vis) {
        float cr, cg, cb;
        float c2r = 0.f, c2g = 0.f, c2b = 0.f;   // second colour set (two-colour walk): precomputed, or the same SH block through a second tone
        bool two = false;                        // wave-uniform
        if constexpr (TONE) two = tone.second != 0;
        if (GEOM_ONLY) {
            cr = cg = cb = 0.0f;      // (sh_colour_kernel writes the three floats and the clamp flags of the Gaussians it colours)
            g.clamped[idx] = 0;
        } else if (p.colors_precomp == nullptr) {
            // computeColorFromSH, forward.cu:20-71
            float dx = px - camx, dy = py - camy, dz = pz - camz;
            const float len = sqrtf(dx * dx + dy * dy + dz * dz);
            dx = dx / len; dy = dy / len; dz = dz / len;
            if (FAST_SH) {
                float sh[48];
#pragma unroll
                for (int q = 0; q < 12; q++) {
                    const float4 v = stage[lane * SH_PITCH4 + q];
                    sh[4 * q] = v.x; sh[4 * q + 1] = v.y; sh[4 * q + 2] = v.z; sh[4 * q + 3] = v.w;
                }
                if constexpr (TONE) {
                    float m[3], o[3], xin, t;
#pragma unroll
                    for (int ch = 0; ch < 3; ch++) {
                        m[ch] = tone.mul ? tone.mul[3 * idx + ch] : 1.0f;
                        o[ch] = tone.offset ? tone.offset[3 * idx + ch] : 0.0f;
                    }
#pragma unroll
                    for (int e = 0; e < 48; e++) sh[e] = tone_value(sh[e], m[e % 3], e < 3 ? o[e] : 0.0f, tone.pre_clamp, tone.post_clamp, xin, t);
                }
                cr = sh_channel(p.D, sh + 0, dx, dy, dz);
                cg = sh_channel(p.D, sh + 1, dx, dy, dz);
                cb = sh_channel(p.D, sh + 2, dx, dy, dz);
                if constexpr (TONE) {
                    if (tone.second) {   // kernel argument: the same coefficients (still staged in LDS) through the second tone
                        float m[3], o[3], xin, t;
#pragma unroll
                        for (int q = 0; q < 12; q++) {
                            const float4 v = stage[lane * SH_PITCH4 + q];
                            sh[4 * q] = v.x; sh[4 * q + 1] = v.y; sh[4 * q + 2] = v.z; sh[4 * q + 3] = v.w;
                        }
#pragma unroll
                        for (int ch = 0; ch < 3; ch++) {
                            m[ch] = tone.mul2 ? tone.mul2[3 * idx + ch] : 1.0f;
                            o[ch] = tone.offset2 ? tone.offset2[3 * idx + ch] : 0.0f;
                        }
#pragma unroll
                        for (int e = 0; e < 48; e++) sh[e] = tone_value(sh[e], m[e % 3], e < 3 ? o[e] : 0.0f, tone.pre_clamp2, tone.post_clamp2, xin, t);
                        c2r = sh_channel(p.D, sh + 0, dx, dy, dz);
                        c2g = sh_channel(p.D, sh + 1, dx, dy, dz);
                        c2b = sh_channel(p.D, sh + 2, dx, dy, dz);
                    }
                }
            } else if constexpr (TONE) {  // generic layout (M != 16 or unaligned): the (D+1)^2 <= 16 coefficients the evaluation reads
                const float* src = p.shs + (size_t)idx * p.M * 3;
                float sh[48], xin, t;
                const int used = 3 * (p.D + 1) * (p.D + 1);
                for (int e = 0; e < 48; e++) {
                    const int ch = e % 3;
                    const float m = tone.mul ? tone.mul[3 * idx + ch] : 1.0f;
                    const float o = (e < 3 && tone.offset) ? tone.offset[3 * idx + ch] : 0.0f;
                    sh[e] = e < used ? tone_value(src[e], m, o, tone.pre_clamp, tone.post_clamp, xin, t) : 0.0f;
                }
                cr = sh_channel(p.D, sh + 0, dx, dy, dz);
                cg = sh_channel(p.D, sh + 1, dx, dy, dz);
                cb = sh_channel(p.D, sh + 2, dx, dy, dz);
                if (tone.second) {
                    for (int e = 0; e < 48; e++) {
                        const int ch = e % 3;
                        const float m = tone.mul2 ? tone.mul2[3 * idx + ch] : 1.0f;
                        const float o = (e < 3 && tone.offset2) ? tone.offset2[3 * idx + ch] : 0.0f;
                        sh[e] = e < used ? tone_value(src[e], m, o, tone.pre_clamp2, tone.post_clamp2, xin, t) : 0.0f;
                    }
                    c2r = sh_channel(p.D, sh + 0, dx, dy, dz);
                    c2g = sh_channel(p.D, sh + 1, dx, dy, dz);
                    c2b = sh_channel(p.D, sh + 2, dx, dy, dz);
                }
            } else {
                const float* sh = p.shs + (size_t)idx * p.M * 3;
                cr = sh_channel(p.D, sh + 0, dx, dy, dz);
                cg = sh_channel(p.D, sh + 1, dx, dy, dz);
                cb = sh_channel(p.D, sh + 2, dx, dy, dz);
            }
            int flags = (cr < 0 ? 1 : 0) | (cg < 0 ? 2 : 0) | (cb < 0 ? 4 : 0);
            if (two) flags |= (c2r < 0 ? 8 : 0) | (c2g < 0 ? 16 : 0) | (c2b < 0 ? 32 : 0);   // (only the TONE instantiations can get here with two set)
            g.clamped[idx] = (unsigned char)flags;
            cr = fmaxf(cr, 0.0f); cg = fmaxf(cg, 0.0f); cb = fmaxf(cb, 0.0f);
            c2r = fmaxf(c2r, 0.0f); c2g = fmaxf(c2g, 0.0f); c2b = fmaxf(c2b, 0.0f);
        } else {
            cr = p.colors_precomp[3 * idx]; cg = p.colors_precomp[3 * idx + 1]; cb = p.colors_precomp[3 * idx + 2];
            if (p.colors_precomp2 != nullptr) {
                two = true;
                c2r = p.colors_precomp2[3 * idx]; c2g = p.colors_precomp2[3 * idx + 1]; c2b = p.colors_precomp2[3 * idx + 2];
            }
        }
        g.depths[idx] = vz;
        // radius 0 with a tile: only a NaN covariance gets here ((int)NaN = 0 in forward.cu:244-246; finite ones have radius >= 2).  The
        // reference counts the tile (tiles_touched = 1, so num_rendered agrees) but its duplicateWithKeys emits nothing for radii <= 0
        // (rasterizer_impl.cu:85) and leaves that list slot UNWRITTEN -- whatever it then composites there is stale memory.  Here the
        // instance exists and draws nothing: its record is parked with opacity 0 and a zero conic (never reached: wg_alpha.h: strip_mask).
        const bool drawn = radius_i > 0;
        conx = drawn ? conx : 0.0f; cony = drawn ? cony : 0.0f; conz = drawn ? conz : 0.0f;
        const float o_eff = drawn ? opacity * coef : 0.0f;
        float4* rec = g.splats + 3 * (size_t)idx;
        rec[0] = make_float4(pixx, pixy, conx, cony);
        if (two) {   // Two-colour walk: the second set rides in the record's three spare floats
            static_assert(WG_STRIP_EXACT == 1, "the box strip test keeps the splat's extent in r2.zw");
            rec[1] = make_float4(conz, o_eff, c2r, cr);
            rec[2] = make_float4(cg, cb, c2g, c2b);
        } else {
            rec[1] = make_float4(conz, o_eff, 0.f, cr);  // .z: spare (the backward kernel parks the Gaussian's id there in LDS)
            const float2 ext = splat_extent(conx, cony, conz, o_eff);  // the box variant of the strip tests (wg_alpha.h)
            rec[2] = make_float4(cg, cb, ext.x, ext.y);
        }
    }
    // always written (all-zero for a culled Gaussian): the binning kernels read the rectangle only
    g.rects[idx] = vis ? make_ushort4((unsigned short)rminx, (unsigned short)rminy, (unsigned short)rmaxx, (unsigned short)rmaxy)
                       : make_ushort4(0, 0, 0, 0);
    g.radii[idx] = radius_i;
    if (radii_out) radii_out[idx] = radius_i;
    g.tiles_touched[idx] = touched;
}

// The colour half of a split frame (GEOM_ONLY above): computeColorFromSH (forward.cu:20-71) with the operations, their order and the
// -ffp-contract=off of preprocess_kernel -- the same bits -- into the three colour floats of the 48-byte record (r1.w, r2.x, r2.y) and `clamped`.
// FAR = false: the Gaussians at or below the frame's near threshold (every visible one when the device switched the split off); FAR = true: the
// others, and only when some tile asked for its far instances (split->need_far, set by the first fix-up phase).  One Gaussian per lane, its
// twelve 16-byte loads at a 192-byte stride: with one lane in ten active a wave touches a tenth of its block's lines (a coalesced block load
// would fetch all of them for one near Gaussian in 64).
template <bool FAR, bool VEC>
__global__ void __launch_bounds__(256) sh_colour_kernel(FwdParams p, GeometryState g, const SplitState* __restrict__ split) {
    const uint32_t near_code = split->near_code;
    if (FAR && (split->need_far == 0u || near_code == SPLIT_OFF)) return;   // uniform: nobody asked
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= p.P || g.tiles_touched[idx] == 0u) return;
    const bool is_near = near_code == SPLIT_OFF || depth_code(__float_as_uint(g.depths[idx]), SPLIT_BITS) <= near_code;
    if (is_near == FAR) return;
    const float camx = p.cam_pos[0], camy = p.cam_pos[1], camz = p.cam_pos[2];
    const float px = p.means3D[3 * idx], py = p.means3D[3 * idx + 1], pz = p.means3D[3 * idx + 2];
    float dx = px - camx, dy = py - camy, dz = pz - camz;
    const float len = sqrtf(dx * dx + dy * dy + dz * dz);
    dx = dx / len; dy = dy / len; dz = dz / len;
    float cr, cg, cb;
    if (VEC) {   // M == 16, 16-byte aligned
        const float4* src = reinterpret_cast<const float4*>(p.shs) + (size_t)idx * 12;
        float sh[48];
#pragma unroll
        for (int q = 0; q < 12; q++) {
            const float4 v = src[q];
            sh[4 * q] = v.x; sh[4 * q + 1] = v.y; sh[4 * q + 2] = v.z; sh[4 * q + 3] = v.w;
        }
        cr = sh_channel(p.D, sh + 0, dx, dy, dz);
        cg = sh_channel(p.D, sh + 1, dx, dy, dz);
        cb = sh_channel(p.D, sh + 2, dx, dy, dz);
    } else {
        const float* sh = p.shs + (size_t)idx * p.M * 3;
        cr = sh_channel(p.D, sh + 0, dx, dy, dz);
        cg = sh_channel(p.D, sh + 1, dx, dy, dz);
        cb = sh_channel(p.D, sh + 2, dx, dy, dz);
    }
    g.clamped[idx] = (unsigned char)((cr < 0 ? 1 : 0) | (cg < 0 ? 2 : 0) | (cb < 0 ? 4 : 0));
    float* rec = reinterpret_cast<float*>(g.splats + 3 * (size_t)idx);
    rec[7] = fmaxf(cr, 0.0f);
    rec[8] = fmaxf(cg, 0.0f);
    rec[9] = fmaxf(cb, 0.0f);
}

hipError_t launch_sh_colour(const FwdParams& p, const GeometryState& g, const SplitState* split, bool far, hipStream_t stream) {
    if (p.P <= 0) return hipSuccess;
    const bool vec = p.M == 16 && (reinterpret_cast<uintptr_t>(p.shs) % 16 == 0);
    const dim3 grid((p.P + 255) / 256), block(256);
    if (far && vec) hipLaunchKernelGGL((sh_colour_kernel<true, true>), grid, block, 0, stream, p, g, split);
    else if (far) hipLaunchKernelGGL((sh_colour_kernel<true, false>), grid, block, 0, stream, p, g, split);
    else if (vec) hipLaunchKernelGGL((sh_colour_kernel<false, true>), grid, block, 0, stream, p, g, split);
    else hipLaunchKernelGGL((sh_colour_kernel<false, false>), grid, block, 0, stream, p, g, split);
    return hipGetLastError();
}

__global__ void __launch_bounds__(256) mark_visible_kernel(int P, const float* __restrict__ means3D,
                                                           const float* __restrict__ vm, unsigned char* __restrict__ present) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= P) return;
    const float px = means3D[3 * idx], py = means3D[3 * idx + 1], pz = means3D[3 * idx + 2];
    const float vz = vm[2] * px + vm[6] * py + vm[10] * pz + vm[14];
    present[idx] = !(vz <= 0.2f);  // auxiliary.h:154
}

hipError_t launch_preprocess(const FwdParams& p, const ShTone& tone_in, const GeometryState& g, int* radii_out, bool geom_only, hipStream_t stream) {
    if (p.P <= 0) return hipSuccess;
    const bool fast = p.shs != nullptr && p.colors_precomp == nullptr && p.M == 16 && (reinterpret_cast<uintptr_t>(p.shs) % 16 == 0);
    const bool pre = p.cov3D_precomp != nullptr;
    const dim3 grid((p.P + 63) / 64), block(64);
    const bool tone = tone_in.enabled && p.shs != nullptr && p.colors_precomp == nullptr;
    if (geom_only) {   // (api.hip asks for it with plain SH colours only; launch_sh_colour supplies the colours)
        if (tone || p.shs == nullptr || p.colors_precomp != nullptr || p.colors_precomp2 != nullptr) return hipErrorInvalidValue;
        if (pre) hipLaunchKernelGGL((preprocess_kernel<0, true, false, true>), grid, block, 0, stream, p, g, radii_out, NoTone{});
        else hipLaunchKernelGGL((preprocess_kernel<0, false, false, true>), grid, block, 0, stream, p, g, radii_out, NoTone{});
        return hipGetLastError();
    }
#define WG_LAUNCH(F, C) hipLaunchKernelGGL((preprocess_kernel<F, C, false>), grid, block, 0, stream, p, g, radii_out, NoTone{})
#define WG_LAUNCH_TONE(F, C) hipLaunchKernelGGL((preprocess_kernel<F, C, true>), grid, block, 0, stream, p, g, radii_out, tone_in)
#define WG_LAUNCH_FAST(L, C) do { if (p.nt_stream) L(2, C); else L(1, C); } while (0)
    if (tone) {
        if (fast && !pre) WG_LAUNCH_FAST(WG_LAUNCH_TONE, false);
        else if (fast) WG_LAUNCH_FAST(WG_LAUNCH_TONE, true);
        else if (!pre) WG_LAUNCH_TONE(0, false);
        else WG_LAUNCH_TONE(0, true);
    } else if (fast && !pre) WG_LAUNCH_FAST(WG_LAUNCH, false);
    else if (fast) WG_LAUNCH_FAST(WG_LAUNCH, true);
    else if (!pre) WG_LAUNCH(0, false);
    else WG_LAUNCH(0, true);
#undef WG_LAUNCH_FAST
#undef WG_LAUNCH
#undef WG_LAUNCH_TONE
    return hipGetLastError();
}

// Geometry reuse (api.hip: wg_forward_args::recolor): a second rasterization of the SAME Gaussians through the same camera with
// other precomputed colours -- WildGaussians renders raw and toned colours over identical geometry, method.py:1573-1611 -- needs none
// of K1's projection and none of the binning again.  This kernel gives the call a geometry state of its own (its backward pass
// accumulates into its own gradient records): everything the backward kernels read is copied from the parent state, the splat
// records with the new colours in place of the old.  One Gaussian per thread, ~200 B per Gaussian of traffic (K1 + binning: the
// whole forward pass in front of the compositing).
__global__ void __launch_bounds__(256) recolor_kernel(int P, GeometryState src, GeometryState dst, const float* __restrict__ colors, int* __restrict__ radii_out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= P) return;
    const int r = src.radii[idx];
    dst.radii[idx] = r;
    if (radii_out) radii_out[idx] = r;
    dst.depths[idx] = src.depths[idx];
    dst.clamped[idx] = 0;  // precomputed colours are never clamped (forward.cu:253-258)
    dst.rects[idx] = src.rects[idx];
    dst.tiles_touched[idx] = src.tiles_touched[idx];
    if (r > 0) {
        const float4* a = src.splats + 3 * (size_t)idx;
        float4* b = dst.splats + 3 * (size_t)idx;
        float4 r0 = a[0], r1 = a[1], r2 = a[2];
        r1.w = colors[3 * idx];
        r2.x = colors[3 * idx + 1];
        r2.y = colors[3 * idx + 2];
        b[0] = r0; b[1] = r1; b[2] = r2;
#pragma unroll
        for (int k = 0; k < 6; k++) dst.cov3D[6 * (size_t)idx + k] = src.cov3D[6 * (size_t)idx + k];
    }
}

hipError_t launch_recolor(int P, const GeometryState& src, const GeometryState& dst, const float* colors, int* radii_out, hipStream_t stream) {
    if (P <= 0) return hipSuccess;
    hipLaunchKernelGGL(recolor_kernel, dim3((P + 255) / 256), dim3(256), 0, stream, P, src, dst, colors, radii_out);
    return hipGetLastError();
}

hipError_t launch_mark_visible(int P, const float* means3D, const float* viewmatrix, unsigned char* present, hipStream_t stream) {
    if (P <= 0) return hipSuccess;
    hipLaunchKernelGGL(mark_visible_kernel, dim3((P + 255) / 256), dim3(256), 0, stream, P, means3D, viewmatrix, present);
    return hipGetLastError();
}

}  // namespace wg
