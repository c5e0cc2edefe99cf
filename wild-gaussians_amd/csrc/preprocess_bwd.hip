// K10 + K11 fused: per-Gaussian backward of the preprocess stage for gfx950.
// Replaces computeCov2DCUDA (backward.cu:144-310), preprocessCUDA backward (backward.cu:382-432), the SH
// backward (backward.cu:20-139) and the covariance backward (backward.cu:314-377), which the reference runs
// as two kernels with dL_dcov3D / dL_dmean3D round-tripping through HBM in between.
//
// Streaming kernel, one Gaussian per lane, one 64-lane wave per workgroup.  The only wide per-Gaussian arrays
// are the SH coefficients in (192 B at M = 16) and their gradients out (192 B): read or written one float per
// lane they are a 192-byte-stride access (64 different cache lines per instruction, 96 such instructions).  With
// M == 16 the wave instead moves its 64 Gaussians' 12 KiB block with 16-byte coalesced accesses and transposes
// it through LDS (row pitch 52 floats = 13 float4: conflict-free ds_read/ds_write_b128 for every 16-lane group).
// Every output except the four accumulation targets of the per-tile pass is fully written here (zeros for culled
// Gaussians), so the caller does not have to clear them.
//
// The formulas are the reference's hand-derived ones, including its quirks:
//  - dL_dopacity arrives w.r.t. the mip-filtered opacity and leaves multiplied by coef (or zeroed when the
//    filter determinant is degenerate), backward.cu:238-248;
//  - clamped t.x/t.y mask only the x/y terms (x_grad_mul / y_grad_mul), backward.cu:180-181,298-300;
//  - dL_dmean3D is OVERWRITTEN by the covariance path then accumulated (backward.cu:309,423,138);
//  - the quaternion gradient is w.r.t. the un-normalised quaternion (backward.cu:376).
#include "wg_common.h"
#include "wg_act.h"

namespace wg {

constexpr float C0 = 0.28209479177387814f;
constexpr float C1 = 0.4886025119029199f;
constexpr float C2a = 1.0925484305920792f, C2b = -1.0925484305920792f, C2c = 0.31539156525252005f, C2d = -1.0925484305920792f,
                C2e = 0.5462742152960396f;
constexpr float C3a = -0.5900435899266435f, C3b = 2.890611442640554f, C3c = -0.4570457994644658f, C3d = 0.3731763325901154f,
                C3e = -0.4570457994644658f, C3f = 1.445305721320277f, C3g = -0.5900435899266435f;

constexpr int SH_PITCH4 = 13;  // float4 per Gaussian in LDS (12 used + 1 pad)

// SH basis values B[k] and their derivatives w.r.t. the (unnormalised-looking) direction components, for the
// polynomial of forward.cu:30-62; coefficients of degrees above `deg` are zeroed.
__device__ __forceinline__ void sh_basis(int deg, float x, float y, float z, float* B, float* Dx, float* Dy, float* Dz) {
    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    const float m1 = deg > 0 ? 1.f : 0.f, m2 = deg > 1 ? 1.f : 0.f, m3 = deg > 2 ? 1.f : 0.f;
    B[0] = C0; Dx[0] = 0.f; Dy[0] = 0.f; Dz[0] = 0.f;
    B[1] = m1 * -C1 * y; Dx[1] = 0.f; Dy[1] = m1 * -C1; Dz[1] = 0.f;
    B[2] = m1 * C1 * z; Dx[2] = 0.f; Dy[2] = 0.f; Dz[2] = m1 * C1;
    B[3] = m1 * -C1 * x; Dx[3] = m1 * -C1; Dy[3] = 0.f; Dz[3] = 0.f;
    B[4] = m2 * C2a * xy; Dx[4] = m2 * C2a * y; Dy[4] = m2 * C2a * x; Dz[4] = 0.f;
    B[5] = m2 * C2b * yz; Dx[5] = 0.f; Dy[5] = m2 * C2b * z; Dz[5] = m2 * C2b * y;
    B[6] = m2 * C2c * (2.f * zz - xx - yy); Dx[6] = m2 * C2c * -2.f * x; Dy[6] = m2 * C2c * -2.f * y; Dz[6] = m2 * C2c * 4.f * z;
    B[7] = m2 * C2d * xz; Dx[7] = m2 * C2d * z; Dy[7] = 0.f; Dz[7] = m2 * C2d * x;
    B[8] = m2 * C2e * (xx - yy); Dx[8] = m2 * C2e * 2.f * x; Dy[8] = m2 * C2e * -2.f * y; Dz[8] = 0.f;
    B[9] = m3 * C3a * y * (3.f * xx - yy); Dx[9] = m3 * C3a * 6.f * xy; Dy[9] = m3 * C3a * 3.f * (xx - yy); Dz[9] = 0.f;
    B[10] = m3 * C3b * xy * z; Dx[10] = m3 * C3b * yz; Dy[10] = m3 * C3b * xz; Dz[10] = m3 * C3b * xy;
    B[11] = m3 * C3c * y * (4.f * zz - xx - yy); Dx[11] = m3 * C3c * -2.f * xy; Dy[11] = m3 * C3c * (-3.f * yy + 4.f * zz - xx);
    Dz[11] = m3 * C3c * 8.f * yz;
    B[12] = m3 * C3d * z * (2.f * zz - 3.f * xx - 3.f * yy); Dx[12] = m3 * C3d * -6.f * xz; Dy[12] = m3 * C3d * -6.f * yz;
    Dz[12] = m3 * C3d * 3.f * (2.f * zz - xx - yy);
    B[13] = m3 * C3e * x * (4.f * zz - xx - yy); Dx[13] = m3 * C3e * (-3.f * xx + 4.f * zz - yy); Dy[13] = m3 * C3e * -2.f * xy;
    Dz[13] = m3 * C3e * 8.f * xz;
    B[14] = m3 * C3f * z * (xx - yy); Dx[14] = m3 * C3f * 2.f * xz; Dy[14] = m3 * C3f * -2.f * yz; Dz[14] = m3 * C3f * (xx - yy);
    B[15] = m3 * C3g * x * (xx - 3.f * yy); Dx[15] = m3 * C3g * 3.f * (xx - yy); Dy[15] = m3 * C3g * -6.f * xy; Dz[15] = 0.f;
}

// Load schedule: the memory counter retires in issue order, so every per-Gaussian input is requested first, the
// 12 KiB SH block second (into registers), and only the SH part of the arithmetic -- placed last -- waits for it.
// RECORD: the per-tile pass left its raw sums in grad_rec (wg_common.h: GRAD_REC_*); this kernel applies the per-Gaussian factors
// and WRITES dL_dmean2D / dL_dconic / dL_dopacity / dL_dcolor for every Gaussian.  !RECORD: those four arrive accumulated.
template <int SH_MODE, bool HAS_SCALES, bool TONE, bool RECORD>   // SH_MODE: 0 generic layout, 1 coalesced blocks through LDS, 2 the same with non-temporal accesses
__global__ void __launch_bounds__(64) preprocess_backward_kernel(
    BwdParams p, const float4* __restrict__ splats, const unsigned char* __restrict__ clamped, const float4* __restrict__ grad_rec,
    float* __restrict__ dL_dmean2D, float* __restrict__ dL_dconic, float* __restrict__ dL_dopacity,
    float* __restrict__ dL_dcolor, float* __restrict__ dL_dmean3D, float* __restrict__ dL_dcov3D,
    float* __restrict__ dL_dsh, float* __restrict__ dL_dscale, float* __restrict__ dL_drot, ToneArg<TONE> tone) {
    constexpr bool FAST_SH = SH_MODE != 0, NT = SH_MODE == 2;
    __shared__ float4 stage[FAST_SH ? 64 * SH_PITCH4 : 1];
    const int lane = threadIdx.x;
    const int base = blockIdx.x * 64;
    const int idx = base + lane;
    const bool in = idx < p.P;
    const int ld = in ? idx : p.P - 1;

    float vm[16], proj[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
        vm[i] = p.viewmatrix[i];
        proj[i] = p.projmatrix[i];
    }
    // the camera position only enters through the SH view direction: absent (null) with precomputed colours, as in the reference
    float camx = 0.f, camy = 0.f, camz = 0.f;
    if (p.shs != nullptr) { camx = p.campos[0]; camy = p.campos[1]; camz = p.campos[2]; }

    // (non-const: they are operands of the ordering fence below, which keeps the compiler from sinking the loads)
    int radius = p.radii[ld];
    float mx = p.means3D[3 * ld], my = p.means3D[3 * ld + 1], mz = p.means3D[3 * ld + 2];
    const float* cv = p.cov3D + 6 * (size_t)ld;
    float v0 = cv[0], v1 = cv[1], v2 = cv[2], v3 = cv[3], v4 = cv[4], v5 = cv[5];
    float4 dconic;
    float combined_opacity = splats[3 * (size_t)ld + 1].y;
    float dLdo_in, g2x, g2y, g2abs = 0.f, dcol0, dcol1, dcol2;
    if (RECORD) {
        const float4 r0 = grad_rec[3 * (size_t)ld], r1 = grad_rec[3 * (size_t)ld + 1], r2 = grad_rec[3 * (size_t)ld + 2];
        dcol0 = r0.x; dcol1 = r0.y; dcol2 = r0.z;
        g2x = r0.w; g2y = r1.x; g2abs = r1.y;                       // raw sums: scaled below, once the loads have landed
        dconic = make_float4(r1.z, r1.w, 0.f, r2.x);
        dLdo_in = r2.y;
    } else {
        dconic = reinterpret_cast<const float4*>(dL_dconic)[ld];
        dLdo_in = dL_dopacity[ld];
        g2x = dL_dmean2D[3 * ld]; g2y = dL_dmean2D[3 * ld + 1];
        dcol0 = dL_dcolor[3 * ld]; dcol1 = dL_dcolor[3 * ld + 1]; dcol2 = dL_dcolor[3 * ld + 2];
    }
    int cl = clamped[ld];
    float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
    float sc0 = 0.f, sc1 = 0.f, sc2 = 0.f;
    float raw_op = 0.f, filt = 0.f;   // raw-parameter mode (wg_raw_gaussians): loaded with the rest, through pointers that are always valid
    if (HAS_SCALES) {
        q = reinterpret_cast<const float4*>(p.rotations)[ld];
        sc0 = p.scales[3 * ld]; sc1 = p.scales[3 * ld + 1]; sc2 = p.scales[3 * ld + 2];
        raw_op = (p.raw_opacities ? p.raw_opacities : p.scales)[ld];
        filt = (p.filter_3D ? p.filter_3D : p.scales)[ld];
    }
    __builtin_amdgcn_sched_barrier(0);  // the machine scheduler would otherwise slip some of the loads above behind the SH ones
    float4 sr0, sr1, sr2, sr3, sr4, sr5, sr6, sr7, sr8, sr9, sr10, sr11;
    sr0 = sr1 = sr2 = sr3 = sr4 = sr5 = sr6 = sr7 = sr8 = sr9 = sr10 = sr11 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (FAST_SH) {  // 64 Gaussians x 12 float4, coalesced
        const float4* src = reinterpret_cast<const float4*>(p.shs) + (size_t)base * 12;
        const int last = min(64, p.P - base) * 12 - 1;
#define WG_SH_LOAD(i) sr##i = stream_load4<NT>(&src[min(i * 64 + lane, last)]);
        WG_SH_LOAD(0) WG_SH_LOAD(1) WG_SH_LOAD(2) WG_SH_LOAD(3) WG_SH_LOAD(4) WG_SH_LOAD(5)
        WG_SH_LOAD(6) WG_SH_LOAD(7) WG_SH_LOAD(8) WG_SH_LOAD(9) WG_SH_LOAD(10) WG_SH_LOAD(11)
#undef WG_SH_LOAD
    }
    // nothing below moves above the SH loads, none of the loads above sinks below them
    asm volatile(""
                 : "+v"(mx), "+v"(my), "+v"(mz), "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(dconic.x), "+v"(dconic.y),
                   "+v"(dconic.w), "+v"(combined_opacity), "+v"(dLdo_in), "+v"(g2x), "+v"(g2y), "+v"(g2abs), "+v"(cl), "+v"(dcol0), "+v"(dcol1),
                   "+v"(dcol2), "+v"(q.x), "+v"(q.y), "+v"(q.z), "+v"(q.w), "+v"(sc0), "+v"(sc1), "+v"(sc2), "+v"(radius), "+v"(raw_op), "+v"(filt)
                 :
                 : "memory");
    // raw-parameter mode: what the forward kernel made of the raw parameters, recomputed (nothing was saved); the gradients of the
    // activated values are turned into those of the raw ones where they are written, below
    const bool raw_mode = HAS_SCALES && p.filter_3D != nullptr;   // wave-uniform
    ActFwd act{};
    if (raw_mode) {
        act = act_forward(q, sc0, sc1, sc2, raw_op, filt);
        q = act.q;
        sc0 = act.sc[0]; sc1 = act.sc[1]; sc2 = act.sc[2];
    }
    const bool vis = in && radius > 0;
    if (RECORD) {
        // the factors the per-tile pass leaves out (render_bwd.hip): mean2D.x = o * 0.5W / log2e * sum(q u'), .y likewise with 0.5H,
        // .z = |o| / log2e * sum(...), conic = -0.5 o * sum(q d d), colour and opacity as summed.  A culled Gaussian's record is
        // zero but its splat record (hence o) was never written: everything is forced to zero there.
        constexpr float INV_L = 1.0f / 1.4426950408889634f;
        const float o = vis ? combined_opacity : 0.f;
        g2x = vis ? g2x * (o * (0.5f * p.W * INV_L)) : 0.f;
        g2y = vis ? g2y * (o * (0.5f * p.H * INV_L)) : 0.f;
        g2abs = vis ? g2abs * (fabsf(o) * INV_L) : 0.f;
        dconic.x = vis ? dconic.x * (o * -0.5f) : 0.f;
        dconic.y = vis ? dconic.y * (o * -0.5f) : 0.f;
        dconic.w = vis ? dconic.w * (o * -0.5f) : 0.f;
        if (!vis) { dcol0 = dcol1 = dcol2 = 0.f; dLdo_in = 0.f; }
    }

    float dcov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float gmx = 0.f, gmy = 0.f, gmz = 0.f;
    float dsc[3] = {0.f, 0.f, 0.f};
    float4 dq = make_float4(0.f, 0.f, 0.f, 0.f);
    float dLdo_out = dLdo_in;
    bool write_dLdo = false;

    if (vis) {
        // ------------------------------------------------------------------ K10: backward.cu:167-310
        const float dcx = dconic.x, dcy = dconic.y, dcz = dconic.w;

        float tx = vm[0] * mx + vm[4] * my + vm[8] * mz + vm[12];
        float ty = vm[1] * mx + vm[5] * my + vm[9] * mz + vm[13];
        const float tz = vm[2] * mx + vm[6] * my + vm[10] * mz + vm[14];
        const float limx = 1.3f * p.tan_fovx, limy = 1.3f * p.tan_fovy;
        const float txtz = tx / tz, tytz = ty / tz;
        tx = fminf(limx, fmaxf(-limx, txtz)) * tz;
        ty = fminf(limy, fmaxf(-limy, tytz)) * tz;
        const float x_grad_mul = (txtz < -limx || txtz > limx) ? 0.f : 1.f;
        const float y_grad_mul = (tytz < -limy || tytz > limy) ? 0.f : 1.f;

        const float hx = p.focal_x, hy = p.focal_y;
        const float j00 = hx / tz, j02 = -(hx * tx) / (tz * tz), j11 = hy / tz, j12 = -(hy * ty) / (tz * tz);
        // T[c][r] (c = column 0/1, r = row): see preprocess.hip
        const float T00 = vm[0] * j00 + vm[2] * j02, T01 = vm[4] * j00 + vm[6] * j02, T02 = vm[8] * j00 + vm[10] * j02;
        const float T10 = vm[1] * j11 + vm[2] * j12, T11 = vm[5] * j11 + vm[6] * j12, T12 = vm[9] * j11 + vm[10] * j12;
        // P0k = sum_r T[0][r] V[k][r], P1k likewise (V symmetric)
        const float P00 = T00 * v0 + T01 * v1 + T02 * v2, P01 = T00 * v1 + T01 * v3 + T02 * v4, P02 = T00 * v2 + T01 * v4 + T02 * v5;
        const float P10 = T10 * v0 + T11 * v1 + T12 * v2, P11 = T10 * v1 + T11 * v3 + T12 * v4, P12 = T10 * v2 + T11 * v4 + T12 * v5;
        const float a0 = P00 * T00 + P01 * T01 + P02 * T02;  // cov2D[0][0] before the filter
        const float b = P10 * T00 + P11 * T01 + P12 * T02;   // cov2D[0][1]
        const float c0 = P10 * T10 + P11 * T11 + P12 * T12;  // cov2D[1][1]
        const float ks = p.kernel_size;

        const float det_0 = fmaxf(1e-6f, a0 * c0 - b * b);
        const float det_1 = fmaxf(1e-6f, (a0 + ks) * (c0 + ks) - b * b);
        const float coef = sqrtf(det_0 / (det_1 + 1e-6f) + 1e-6f);
        const bool degenerate = (det_0 <= 1e-6f) || (det_1 <= 1e-6f);

        const float opacity = combined_opacity / (coef + 1e-6f);
        const float dL_dcoef = dLdo_in * opacity;
        const float dL_dsqrtcoef = dL_dcoef * 0.5f / (coef + 1e-6f);
        const float dL_ddet0 = dL_dsqrtcoef / (det_1 + 1e-6f);
        const float dL_ddet1 = dL_dsqrtcoef * det_0 * (-1.f / (det_1 * det_1 + 1e-6f));
        const float dcoef_da = dL_ddet0 * c0 + dL_ddet1 * (c0 + ks);
        const float dcoef_db = (dL_ddet0 + dL_ddet1) * (-2.f * b);
        const float dcoef_dc = dL_ddet0 * a0 + dL_ddet1 * (a0 + ks);

        const float a = a0 + ks, c = c0 + ks;
        const float denom = a * c - b * b;
        float dL_da = 0.f, dL_db = 0.f, dL_dc = 0.f;
        const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
        if (denom2inv != 0.f) {
            dL_da = denom2inv * (-c * c * dcx + 2 * b * c * dcy + (denom - a * c) * dcz);
            dL_dc = denom2inv * (-a * a * dcz + 2 * a * b * dcy + (denom - a * c) * dcx);
            dL_db = denom2inv * 2 * (b * c * dcx - (denom + 2 * b * b) * dcy + a * b * dcz);
            write_dLdo = true;
            if (degenerate) {
                dLdo_out = 0.f;
            } else {
                dL_da += dcoef_da;
                dL_dc += dcoef_dc;
                dL_db += dcoef_db;
                dLdo_out = dLdo_in * coef;
            }
            dcov[0] = T00 * T00 * dL_da + T00 * T10 * dL_db + T10 * T10 * dL_dc;
            dcov[3] = T01 * T01 * dL_da + T01 * T11 * dL_db + T11 * T11 * dL_dc;
            dcov[5] = T02 * T02 * dL_da + T02 * T12 * dL_db + T12 * T12 * dL_dc;
            dcov[1] = 2 * T00 * T01 * dL_da + (T00 * T11 + T01 * T10) * dL_db + 2 * T10 * T11 * dL_dc;
            dcov[2] = 2 * T00 * T02 * dL_da + (T00 * T12 + T02 * T10) * dL_db + 2 * T10 * T12 * dL_dc;
            dcov[4] = 2 * T02 * T01 * dL_da + (T01 * T12 + T02 * T11) * dL_db + 2 * T11 * T12 * dL_dc;
        }

        // dL/dT (backward.cu:273-284), dL/dJ (:288-291), dL/dt (:293-300)
        const float dT00 = 2 * P00 * dL_da + P10 * dL_db, dT01 = 2 * P01 * dL_da + P11 * dL_db, dT02 = 2 * P02 * dL_da + P12 * dL_db;
        const float dT10 = 2 * P10 * dL_dc + P00 * dL_db, dT11 = 2 * P11 * dL_dc + P01 * dL_db, dT12 = 2 * P12 * dL_dc + P02 * dL_db;
        const float dJ00 = vm[0] * dT00 + vm[4] * dT01 + vm[8] * dT02;
        const float dJ02 = vm[2] * dT00 + vm[6] * dT01 + vm[10] * dT02;
        const float dJ11 = vm[1] * dT10 + vm[5] * dT11 + vm[9] * dT12;
        const float dJ12 = vm[2] * dT10 + vm[6] * dT11 + vm[10] * dT12;
        const float itz = 1.f / tz, itz2 = itz * itz, itz3 = itz2 * itz;
        const float dtx = x_grad_mul * -hx * itz2 * dJ02;
        const float dty = y_grad_mul * -hy * itz2 * dJ12;
        const float dtz = -hx * itz2 * dJ00 - hy * itz2 * dJ11 + (2 * hx * tx) * itz3 * dJ02 + (2 * hy * ty) * itz3 * dJ12;
        // transformVec4x3Transpose, auxiliary.h:89-97
        gmx = vm[0] * dtx + vm[1] * dty + vm[2] * dtz;
        gmy = vm[4] * dtx + vm[5] * dty + vm[6] * dtz;
        gmz = vm[8] * dtx + vm[9] * dty + vm[10] * dtz;

        // ------------------------------------------------------------------ K11: backward.cu:406-423
        {
            const float hw = proj[3] * mx + proj[7] * my + proj[11] * mz + proj[15];
            const float m_w = 1.0f / (hw + 0.0000001f);
            const float mul1 = (proj[0] * mx + proj[4] * my + proj[8] * mz + proj[12]) * m_w * m_w;
            const float mul2 = (proj[1] * mx + proj[5] * my + proj[9] * mz + proj[13]) * m_w * m_w;
            gmx += (proj[0] * m_w - proj[3] * mul1) * g2x + (proj[1] * m_w - proj[3] * mul2) * g2y;
            gmy += (proj[4] * m_w - proj[7] * mul1) * g2x + (proj[5] * m_w - proj[7] * mul2) * g2y;
            gmz += (proj[8] * m_w - proj[11] * mul1) * g2x + (proj[9] * m_w - proj[11] * mul2) * g2y;
        }

        // ------------------------------------------------------------------ covariance backward, backward.cu:314-377
        if (HAS_SCALES) {
            const float r = q.x, x = q.y, y = q.z, z = q.w;
            const float s[3] = {p.scale_modifier * sc0, p.scale_modifier * sc1, p.scale_modifier * sc2};
            // R[c][r] column-major as filled by the reference
            const float R[3][3] = {{1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y)},
                                   {2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x)},
                                   {2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y)}};
            // dL_dSigma (symmetric, off-diagonals halved), dS[c][k]
            const float dS[3][3] = {{dcov[0], 0.5f * dcov[1], 0.5f * dcov[2]},
                                    {0.5f * dcov[1], dcov[3], 0.5f * dcov[4]},
                                    {0.5f * dcov[2], 0.5f * dcov[4], dcov[5]}};
            // dL_dM[c][w] = sum_k 2*M[k][w]*dS[c][k],  M[k][w] = s_w * R[k][w]
            float dM[3][3];
#pragma unroll
            for (int cc = 0; cc < 3; cc++)
#pragma unroll
                for (int w = 0; w < 3; w++)
                    dM[cc][w] = 2.f * s[w] * (R[0][w] * dS[cc][0] + R[1][w] * dS[cc][1] + R[2][w] * dS[cc][2]);
            // dL_dscale_w = sum_c R[c][w] * dM[c][w]
#pragma unroll
            for (int w = 0; w < 3; w++) dsc[w] = R[0][w] * dM[0][w] + R[1][w] * dM[1][w] + R[2][w] * dM[2][w];
            // D(c,r) = dL_dMt[c][r] * s_c = dM[r][c] * s_c
#define D(c_, r_) (dM[r_][c_] * s[c_])
            dq.x = 2 * z * (D(0, 1) - D(1, 0)) + 2 * y * (D(2, 0) - D(0, 2)) + 2 * x * (D(1, 2) - D(2, 1));
            dq.y = 2 * y * (D(1, 0) + D(0, 1)) + 2 * z * (D(2, 0) + D(0, 2)) + 2 * r * (D(1, 2) - D(2, 1)) - 4 * x * (D(2, 2) + D(1, 1));
            dq.z = 2 * x * (D(1, 0) + D(0, 1)) + 2 * r * (D(2, 0) - D(0, 2)) + 2 * z * (D(1, 2) + D(2, 1)) - 4 * y * (D(2, 2) + D(0, 0));
            dq.w = 2 * r * (D(0, 1) - D(1, 0)) + 2 * x * (D(2, 0) + D(0, 2)) + 2 * y * (D(1, 2) + D(2, 1)) - 4 * z * (D(1, 1) + D(0, 0));
#undef D
        }
    }

    if (raw_mode) {   // (RECORD only: api.hip refuses the combination otherwise)
        float gs[3], go;
        float4 gr;
        act_backward(act, dq, dsc, write_dLdo ? dLdo_out : dLdo_in, gr, gs, go);
        dq = gr;
        dsc[0] = gs[0]; dsc[1] = gs[1]; dsc[2] = gs[2];
        dLdo_out = dLdo_in = go;
    }

    // ---- the outputs that do not depend on the SH block ----
    if (in) {
        float* o = dL_dcov3D + 6 * (size_t)idx;
        o[0] = dcov[0]; o[1] = dcov[1]; o[2] = dcov[2]; o[3] = dcov[3]; o[4] = dcov[4]; o[5] = dcov[5];
        if (RECORD) {
            dL_dopacity[idx] = write_dLdo ? dLdo_out : dLdo_in;
            dL_dmean2D[3 * idx] = g2x; dL_dmean2D[3 * idx + 1] = g2y; dL_dmean2D[3 * idx + 2] = g2abs;
            // the reference's intermediates: only a caller that wants them passes the pointers (wg_rasterizer.h)
            if (dL_dconic) reinterpret_cast<float4*>(dL_dconic)[idx] = dconic;
            if (dL_dcolor) { dL_dcolor[3 * idx] = dcol0; dL_dcolor[3 * idx + 1] = dcol1; dL_dcolor[3 * idx + 2] = dcol2; }
            if (p.dL_dcolor2) {   // two-colour walk (render_bwd.hip: DUAL): the record's floats 10, 11 and grad_aux; zeros for a culled Gaussian
                const float4 r2 = grad_rec[3 * (size_t)idx + 2];
                const float* aux = reinterpret_cast<const float*>(grad_rec) + (size_t)p.P * GRAD_REC_FLOATS;
                p.dL_dcolor2[3 * idx] = r2.z; p.dL_dcolor2[3 * idx + 1] = r2.w; p.dL_dcolor2[3 * idx + 2] = aux[idx];
            }
        } else if (write_dLdo) {
            dL_dopacity[idx] = dLdo_out;
        }
        if (HAS_SCALES) {
            dL_dscale[3 * idx] = dsc[0];
            dL_dscale[3 * idx + 1] = dsc[1];
            dL_dscale[3 * idx + 2] = dsc[2];
            reinterpret_cast<float4*>(dL_drot)[idx] = dq;
        }
    }

    asm volatile("" : "+v"(gmx), "+v"(gmy), "+v"(gmz) : : "memory");  // the SH part stays below the geometry part
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

    // ------------------------------------------------------------------ SH backward, backward.cu:20-139
    float tdm[3] = {0.f, 0.f, 0.f}, tdo[3] = {0.f, 0.f, 0.f};  // TONE: dL/dmul, dL/doffset of this Gaussian
    float tdm2[3] = {0.f, 0.f, 0.f}, tdo2[3] = {0.f, 0.f, 0.f};  // ... of the second tone (ShTone::second)
    float dsh[FAST_SH ? 48 : 1];
    if (FAST_SH) {
#pragma unroll
        for (int k = 0; k < 48; k++) dsh[k] = 0.f;
    }
    if (vis && p.shs != nullptr) {
        const float ox = mx - camx, oy = my - camy, oz = mz - camz;
        const float sum2 = ox * ox + oy * oy + oz * oz;
        const float ilen = 1.0f / sqrtf(sum2);
        float B[16], Dx[16], Dy[16], Dz[16];
        sh_basis(p.D, ox * ilen, oy * ilen, oz * ilen, B, Dx, Dy, Dz);
        const float dRGB[3] = {(cl & 1) ? 0.f : dcol0, (cl & 2) ? 0.f : dcol1, (cl & 4) ? 0.f : dcol2};
        float ddx = 0.f, ddy = 0.f, ddz = 0.f;  // dL_ddir
        // TONE with a second set (ShTone::second): the same coefficients fed a second colour through tone 2; its dL/dRGB are the record's
        // floats 10, 11 and grad_aux (render_bwd.hip: DUAL), its clamp flags bits 3-5.  Both chains add into dL_dsh and dL_ddir.
        float dRGB2[3] = {0.f, 0.f, 0.f}, tm2[3] = {1.f, 1.f, 1.f}, to2[3] = {0.f, 0.f, 0.f};
        bool second = false;
        if constexpr (TONE && RECORD) {
            if (tone.second) {
                second = true;
                const float4 r2 = grad_rec[3 * (size_t)ld + 2];
                const float* aux = reinterpret_cast<const float*>(grad_rec) + (size_t)p.P * GRAD_REC_FLOATS;
                dRGB2[0] = (cl & 8) ? 0.f : r2.z; dRGB2[1] = (cl & 16) ? 0.f : r2.w; dRGB2[2] = (cl & 32) ? 0.f : aux[ld];
#pragma unroll
                for (int ch = 0; ch < 3; ch++) {
                    if (tone.mul2) tm2[ch] = tone.mul2[3 * idx + ch];
                    if (tone.offset2) to2[ch] = tone.offset2[3 * idx + ch];
                }
            }
        }
        // TONE (wg_common.h: ShTone): the evaluation saw min(min(raw, pre) * mul + offset[k == 0], post); chain rule back to the raw
        // coefficients, the multiplier and the offset (clamp_max passes the gradient where x <= max, as torch does)
        float tm[3] = {1.f, 1.f, 1.f}, to[3] = {0.f, 0.f, 0.f};
        if constexpr (TONE) {
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
                if (tone.mul) tm[ch] = tone.mul[3 * idx + ch];
                if (tone.offset) to[ch] = tone.offset[3 * idx + ch];
            }
        }
        if (FAST_SH) {
            float sh[48];
#pragma unroll
            for (int qd = 0; qd < 12; qd++) {
                const float4 v = stage[lane * SH_PITCH4 + qd];
                sh[4 * qd] = v.x; sh[4 * qd + 1] = v.y; sh[4 * qd + 2] = v.z; sh[4 * qd + 3] = v.w;
            }
#pragma unroll
            for (int k = 0; k < 16; k++)
#pragma unroll
                for (int ch = 0; ch < 3; ch++) {
                    float g = B[k] * dRGB[ch], val = sh[3 * k + ch];
                    if constexpr (TONE) {
                        float xin, t;
                        const float raw = val;
                        val = tone_value(raw, tm[ch], k == 0 ? to[ch] : 0.0f, tone.pre_clamp, tone.post_clamp, xin, t);
                        g = (t <= tone.post_clamp) ? g : 0.0f;
                        tdm[ch] += g * xin;
                        if (k == 0) tdo[ch] = g;
                        g = (raw <= tone.pre_clamp) ? g * tm[ch] : 0.0f;
                    }
                    dsh[3 * k + ch] = g;
                    const float w = val * dRGB[ch];
                    ddx += Dx[k] * w;
                    ddy += Dy[k] * w;
                    ddz += Dz[k] * w;
                }
            if constexpr (TONE) {
                if (second) {
#pragma unroll
                    for (int k = 0; k < 16; k++)
#pragma unroll
                        for (int ch = 0; ch < 3; ch++) {
                            float g = B[k] * dRGB2[ch], xin, t;
                            const float raw = sh[3 * k + ch];
                            const float val = tone_value(raw, tm2[ch], k == 0 ? to2[ch] : 0.0f, tone.pre_clamp2, tone.post_clamp2, xin, t);
                            g = (t <= tone.post_clamp2) ? g : 0.0f;
                            tdm2[ch] += g * xin;
                            if (k == 0) tdo2[ch] = g;
                            g = (raw <= tone.pre_clamp2) ? g * tm2[ch] : 0.0f;
                            dsh[3 * k + ch] += g;
                            const float w = val * dRGB2[ch];
                            ddx += Dx[k] * w;
                            ddy += Dy[k] * w;
                            ddz += Dz[k] * w;
                        }
                }
            }
        } else {
            const float* sh = p.shs + (size_t)idx * p.M * 3;
            float* d = dL_dsh + (size_t)idx * p.M * 3;
            const int ncoef = (p.D + 1) * (p.D + 1);
            for (int k = 0; k < p.M; k++)
                for (int ch = 0; ch < 3; ch++) {
                    float bk = 0.f, dxk = 0.f, dyk = 0.f, dzk = 0.f;
                    if (k < ncoef && k < 16) { bk = B[k]; dxk = Dx[k]; dyk = Dy[k]; dzk = Dz[k]; }
                    float g = bk * dRGB[ch], val = sh[3 * k + ch];
                    if constexpr (TONE) {
                        float xin, t;
                        const float raw = val;
                        val = tone_value(raw, tm[ch], k == 0 ? to[ch] : 0.0f, tone.pre_clamp, tone.post_clamp, xin, t);
                        g = (t <= tone.post_clamp) ? g : 0.0f;
                        tdm[ch] += g * xin;
                        if (k == 0) tdo[ch] = g;
                        g = (raw <= tone.pre_clamp) ? g * tm[ch] : 0.0f;
                    }
                    float w = val * dRGB[ch];
                    if constexpr (TONE) {
                        if (second) {
                            float g2 = bk * dRGB2[ch], xin, t;
                            const float raw = sh[3 * k + ch];
                            const float val2 = tone_value(raw, tm2[ch], k == 0 ? to2[ch] : 0.0f, tone.pre_clamp2, tone.post_clamp2, xin, t);
                            g2 = (t <= tone.post_clamp2) ? g2 : 0.0f;
                            tdm2[ch] += g2 * xin;
                            if (k == 0) tdo2[ch] = g2;
                            g += (raw <= tone.pre_clamp2) ? g2 * tm2[ch] : 0.0f;
                            w += val2 * dRGB2[ch];
                        }
                    }
                    d[3 * k + ch] = g;
                    ddx += dxk * w;
                    ddy += dyk * w;
                    ddz += dzk * w;
                }
        }
        // dnormvdv, auxiliary.h:107-117
        const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
        gmx += ((sum2 - ox * ox) * ddx - oy * ox * ddy - oz * ox * ddz) * invsum32;
        gmy += (-ox * oy * ddx + (sum2 - oy * oy) * ddy - oz * oy * ddz) * invsum32;
        gmz += (-ox * oz * ddx - oy * oz * ddy + (sum2 - oz * oz) * ddz) * invsum32;
    }

    // ---- outputs: written for every Gaussian of the range (zeros when culled) ----
    if (in) {
        dL_dmean3D[3 * idx] = gmx;
        dL_dmean3D[3 * idx + 1] = gmy;
        dL_dmean3D[3 * idx + 2] = gmz;
        if (!FAST_SH && p.shs != nullptr && !vis) {
            float* d = dL_dsh + (size_t)idx * p.M * 3;
            for (int k = 0; k < p.M * 3; k++) d[k] = 0.f;
        }
        if constexpr (TONE) {
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
                if (tone.dL_dmul) tone.dL_dmul[3 * idx + ch] = tdm[ch];
                if (tone.dL_doffset) tone.dL_doffset[3 * idx + ch] = tdo[ch];
                if (tone.dL_dmul2) tone.dL_dmul2[3 * idx + ch] = tdm2[ch];
                if (tone.dL_doffset2) tone.dL_doffset2[3 * idx + ch] = tdo2[ch];
            }
        }
    }
    if (FAST_SH) {
        __syncthreads();  // every lane has consumed its SH inputs
#pragma unroll
        for (int qd = 0; qd < 12; qd++)
            stage[lane * SH_PITCH4 + qd] = make_float4(dsh[4 * qd], dsh[4 * qd + 1], dsh[4 * qd + 2], dsh[4 * qd + 3]);
        __syncthreads();
        float4* dst = reinterpret_cast<float4*>(dL_dsh) + (size_t)base * 12;
        const int nvalid = min(64, p.P - base) * 12;
#pragma unroll
        for (int i = 0; i < 12; i++) {
            const int f = i * 64 + lane;
            if (f < nvalid) stream_store4<NT>(&dst[f], stage[(f / 12) * SH_PITCH4 + (f % 12)]);
        }
    }
}

hipError_t launch_preprocess_backward(const BwdParams& p, const ShTone& tone_in, const GeometryState& g, float* dL_dmean2D,
                                      float* dL_dconic, float* dL_dopacity, float* dL_dcolor,
                                      float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale,
                                      float* dL_drot, bool record, hipStream_t stream) {
    if (p.P <= 0) return hipSuccess;
    const dim3 grid((p.P + 63) / 64), block(64);
    const bool fast = p.shs != nullptr && p.M == 16 && (reinterpret_cast<uintptr_t>(p.shs) % 16 == 0) &&
                      (reinterpret_cast<uintptr_t>(dL_dsh) % 16 == 0);
    const bool sc = p.scales != nullptr;
    const bool tone = tone_in.enabled && p.shs != nullptr;
    const float4* rec = reinterpret_cast<const float4*>(g.grad_rec);
#define WG_ARGS p, g.splats, g.clamped, rec, dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale, dL_drot
#define WG_LAUNCH(F, S)                                                                                                              \
    do {                                                                                                                             \
        if (tone && record) hipLaunchKernelGGL((preprocess_backward_kernel<F, S, true, true>), grid, block, 0, stream, WG_ARGS, tone_in);   \
        else if (tone) hipLaunchKernelGGL((preprocess_backward_kernel<F, S, true, false>), grid, block, 0, stream, WG_ARGS, tone_in);       \
        else if (record) hipLaunchKernelGGL((preprocess_backward_kernel<F, S, false, true>), grid, block, 0, stream, WG_ARGS, NoTone{});    \
        else hipLaunchKernelGGL((preprocess_backward_kernel<F, S, false, false>), grid, block, 0, stream, WG_ARGS, NoTone{});               \
    } while (0)
    if (fast && sc) { if (p.nt_stream) WG_LAUNCH(2, true); else WG_LAUNCH(1, true); }
    else if (fast) { if (p.nt_stream) WG_LAUNCH(2, false); else WG_LAUNCH(1, false); }
    else if (sc) WG_LAUNCH(0, true);
    else WG_LAUNCH(0, false);
#undef WG_LAUNCH
#undef WG_ARGS
    return hipGetLastError();
}

}  // namespace wg
