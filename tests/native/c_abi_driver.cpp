// Torch-free driver of the C-ABI (include/wg_rasterizer.h): one forward + backward + markVisible on a small synthetic scene
// with plain hipMalloc buffers and hipMalloc-backed allocator callbacks.  Two uses:
//   * scripts/asan_pass.sh builds it and the library with -fsanitize=address (host side) and runs it on the GPU: the sanitizer
//     pass SURVEY.md section 5 asks for, without a Python interpreter between ASan and the HIP runtime;
//   * evidence that the boundary really is "plain pointers and sizes, no torch types" (tests/test_native_driver.py).
// Since round 5 everything beyond the reference goes through wg_rasterize_{forward,backward}_ex (one struct, optional blocks, per-call
// options): the two-colour, raw-parameter, toned, two-tone and recolouring calls are driven through it and their images checked, bit for bit, against plain
// (toned) calls; the deterministic backward runs three times over one frame (bit-identical) beside the atomic one; three host threads then
// call forward + backward concurrently on their own streams.
// Prints one line "ok num_rendered=... checksum=..." and exits 0, or a diagnostic and a non-zero code.
// With a fourth argument (a path) it also DUMPS its inputs and every output of the three calls there, raw little-endian:
//   int32 {P, W, H, D, M, R}, float32 {tanx, tany}, then float32 arrays means[3P] scales[3P] rots[4P] opac[P] shs[3MP] view[16] proj[16]
//   campos[3] bg[3] cot[3WH] | color[3WH] g2d[3P] gcon[4P] gop[P] gcol[3P] g3d[3P] gcov[6P] gsh[3MP] gsc[3P] grot[4P], int32 radii[P],
//   uint8 vis[P] -- tests/test_native_driver.py feeds the same inputs to the CPU oracle and compares everything.
#include <hip/hip_runtime.h>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <mutex>
#include <thread>
#include <utility>
#include <vector>
#include "wg_rasterizer.h"
#include "wg_activations.h"

#define CHECK_HIP(x)                                                                  \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } \
    } while (0)

struct Grow {  // one scratch buffer, grown on demand like the reference's resizeFunctional (rasterize_points.cu:27-33)
    char* p = nullptr;
    size_t cap = 0;
    static char* alloc(size_t n, void* user) {
        Grow* g = static_cast<Grow*>(user);
        if (n > g->cap) {
            if (g->p) (void)hipFree(g->p);
            g->p = nullptr;
            if (hipMalloc(reinterpret_cast<void**>(&g->p), n ? n : 1) != hipSuccess) return nullptr;
            g->cap = n;
        }
        return g->p;
    }
};

static uint32_t lcg(uint32_t& s) { s = s * 1664525u + 1013904223u; return s; }
static float uni(uint32_t& s) { return (lcg(s) >> 8) * (1.0f / 16777216.0f); }

template <typename T>
static int upload(const std::vector<T>& h, T** d) {
    CHECK_HIP(hipMalloc(reinterpret_cast<void**>(d), h.size() * sizeof(T)));
    CHECK_HIP(hipMemcpy(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return 0;
}

int main(int argc, char** argv) {
    const int P = argc > 1 ? std::atoi(argv[1]) : 20000, W = argc > 2 ? std::atoi(argv[2]) : 320, H = argc > 3 ? std::atoi(argv[3]) : 200;
    const int D = 1, M = 4;
    uint32_t seed = 12345u;
    const float tanx = std::tan(0.5f * 1.0471976f), tany = tanx * H / W;
    std::vector<float> means(3 * P), scales(3 * P), rots(4 * P), opac(P), shs((size_t)P * M * 3);
    for (int i = 0; i < P; i++) {
        const float z = 1.0f + 9.0f * uni(seed);
        means[3 * i] = z * tanx * (2.2f * uni(seed) - 1.1f);
        means[3 * i + 1] = z * tany * (2.2f * uni(seed) - 1.1f);
        means[3 * i + 2] = z;
        for (int k = 0; k < 3; k++) scales[3 * i + k] = 0.02f * z * (0.5f + uni(seed));
        float q[4], n = 0.f;
        for (int k = 0; k < 4; k++) { q[k] = uni(seed) - 0.5f; n += q[k] * q[k]; }
        n = 1.0f / std::sqrt(n + 1e-12f);
        for (int k = 0; k < 4; k++) rots[4 * i + k] = q[k] * n;
        opac[i] = 0.05f + 0.9f * uni(seed);
        for (int k = 0; k < M * 3; k++) shs[(size_t)i * M * 3 + k] = (uni(seed) - 0.5f) * (k < 3 ? 1.0f : 0.2f);
    }
    // camera at the origin looking down +z: view = identity; proj = OpenCV projection (method.py:605-616), both transposed
    const float fx = 0.5f * W / tanx, fy = fx, zn = 0.01f, zf = 100.0f;
    std::vector<float> view = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    std::vector<float> Pm(16, 0.f);  // row-major P, then transposed into proj
    Pm[0] = 2.f * fx / W; Pm[5] = 2.f * fy / H; Pm[10] = zf / (zf - zn); Pm[11] = -(zf * zn) / (zf - zn); Pm[14] = 1.f;
    std::vector<float> proj(16);
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) proj[4 * c + r] = Pm[4 * r + c];
    std::vector<float> campos = {0, 0, 0}, bg = {0.1f, 0.2f, 0.3f};
    std::vector<float> cot((size_t)3 * W * H);
    for (auto& v : cot) v = (uni(seed) - 0.5f) / (3.0f * W * H);

    float *d_means, *d_scales, *d_rots, *d_opac, *d_shs, *d_view, *d_proj, *d_campos, *d_bg, *d_cot;
    if (upload(means, &d_means) || upload(scales, &d_scales) || upload(rots, &d_rots) || upload(opac, &d_opac) || upload(shs, &d_shs) ||
        upload(view, &d_view) || upload(proj, &d_proj) || upload(campos, &d_campos) || upload(bg, &d_bg) || upload(cot, &d_cot))
        return 2;
    float* d_color;
    int* d_radii;
    CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&d_color), (size_t)3 * W * H * sizeof(float)));
    CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&d_radii), (size_t)P * sizeof(int)));
    hipStream_t stream;
    CHECK_HIP(hipStreamCreate(&stream));
    Grow geom, bin, img;

    // argument validation comes before any device work
    if (wg_rasterize_forward(nullptr, nullptr, Grow::alloc, &bin, Grow::alloc, &img, P, D, M, d_bg, W, H, d_means, d_shs, nullptr, d_opac, d_scales,
                             1.0f, d_rots, nullptr, d_view, d_proj, d_campos, tanx, tany, 0.1f, nullptr, 0, d_color, d_radii, 0, stream) !=
        WG_ERR_INVALID_ARGUMENT) {
        std::fprintf(stderr, "a NULL allocator was not refused\n");
        return 3;
    }
    const int R = wg_rasterize_forward(Grow::alloc, &geom, Grow::alloc, &bin, Grow::alloc, &img, P, D, M, d_bg, W, H, d_means, d_shs, nullptr, d_opac,
                                       d_scales, 1.0f, d_rots, nullptr, d_view, d_proj, d_campos, tanx, tany, 0.1f, nullptr, 0, d_color, d_radii,
                                       0, stream);
    if (R <= 0) { std::fprintf(stderr, "forward: %s (%s)\n", wg_status_string(R), wg_last_hip_error()); return 4; }
    // twice more: from the second call on the library speculates on the frame's size from this thread's history (the binning buffer is
    // requested before the count is known, possibly a second time after it); the third with a tiny margin so that Grow is asked to grow
    for (int rep = 0; rep < 2; rep++) {
        if (rep == 1) (void)wg_set_option("spec_margin_pct", 0);
        const int Rr = wg_rasterize_forward(Grow::alloc, &geom, Grow::alloc, &bin, Grow::alloc, &img, P, D, M, d_bg, W, H, d_means, d_shs, nullptr, d_opac,
                                            d_scales, 1.0f, d_rots, nullptr, d_view, d_proj, d_campos, tanx, tany, 0.1f, nullptr, 0, d_color, d_radii,
                                            0, stream);
        if (Rr != R) { std::fprintf(stderr, "repeated forward: %d instead of %d (%s)\n", Rr, R, wg_last_hip_error()); return 4; }
    }
    if (wg_get_option("spec_frames") < 2 || wg_get_option("spec_misses") != 0) {
        std::fprintf(stderr, "speculation did not engage: %d frames, %d misses\n", wg_get_option("spec_frames"), wg_get_option("spec_misses"));
        return 4;
    }
    (void)wg_set_option("spec_margin_pct", 25);

    float *g2d, *gcon, *gop, *gcol, *g3d, *gcov, *gsh, *gsc, *grot;
    CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&g2d), (size_t)P * 3 * 4));
    CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&gcon), (size_t)P * 4 * 4));
    CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&gop), (size_t)P * 4));
    CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&gcol), (size_t)P * 3 * 4));
    CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&g3d), (size_t)P * 3 * 4));
    CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&gcov), (size_t)P * 6 * 4));
    CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&gsh), (size_t)P * M * 3 * 4));
    CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&gsc), (size_t)P * 3 * 4));
    CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&grot), (size_t)P * 4 * 4));
    const int st = wg_rasterize_backward(P, D, M, R, d_bg, W, H, d_means, d_shs, nullptr, d_scales, 1.0f, d_rots, nullptr, d_view, d_proj, d_campos, tanx,
                                         tany, 0.1f, nullptr, d_radii, geom.p, bin.p, img.p, d_cot, g2d, gcon, gop, gcol, g3d, gcov, gsh, gsc, grot, 0,
                                         stream);
    if (st != WG_OK) { std::fprintf(stderr, "backward: %s (%s)\n", wg_status_string(st), wg_last_hip_error()); return 5; }
    // ---- the struct entry points (wg_rasterize_forward_ex / _backward_ex): the reference-shaped arguments once, blocks added per call ----
    auto fwd_args = [&](Grow* g, Grow* b, Grow* i, int deg, int m, const float* shs_, const float* cols, const float* op, const float* sc, const float* rt,
                        float* out, int* rad) {
        wg_forward_args a{};
        a.struct_size = sizeof(a);
        a.geometry_alloc = Grow::alloc; a.geometry_user = g; a.binning_alloc = Grow::alloc; a.binning_user = b; a.image_alloc = Grow::alloc; a.image_user = i;
        a.P = P; a.D = deg; a.M = m; a.width = W; a.height = H;
        a.scale_modifier = 1.0f; a.tan_fovx = tanx; a.tan_fovy = tany; a.kernel_size = 0.1f;
        a.background = d_bg; a.means3D = d_means; a.shs = shs_; a.colors_precomp = cols; a.opacities = op; a.scales = sc; a.rotations = rt;
        a.viewmatrix = d_view; a.projmatrix = d_proj; a.cam_pos = d_campos;
        a.out_color = out; a.radii = rad; a.stream = stream;
        return a;
    };
    auto bwd_args = [&](int deg, int m, int R_, const float* shs_, const float* cols, const float* sc, const float* rt, char* g, char* b, char* i, float* dconic,
                        float* dcol, float* dsh) {
        wg_backward_args a{};
        a.struct_size = sizeof(a);
        a.P = P; a.D = deg; a.M = m; a.R = R_; a.width = W; a.height = H;
        a.scale_modifier = 1.0f; a.tan_fovx = tanx; a.tan_fovy = tany; a.kernel_size = 0.1f;
        a.background = d_bg; a.means3D = d_means; a.shs = shs_; a.colors_precomp = cols; a.scales = sc; a.rotations = rt;
        a.viewmatrix = d_view; a.projmatrix = d_proj; a.campos = d_campos; a.radii = d_radii;
        a.geom_buffer = g; a.binning_buffer = b; a.image_buffer = i; a.dL_dpix = d_cot;
        a.dL_dmean2D = g2d; a.dL_dconic = dconic; a.dL_dopacity = gop; a.dL_dcolor = dcol; a.dL_dmean3D = g3d; a.dL_dcov3D = gcov; a.dL_dsh = dsh;
        a.dL_dscale = gsc; a.dL_drot = grot; a.stream = stream;
        return a;
    };
    {   // a struct of another size (an older / newer header) is refused, and so is a per-call option that contradicts the frame's forward call
        wg_forward_args bad = fwd_args(&geom, &bin, &img, D, M, d_shs, nullptr, d_opac, d_scales, d_rots, d_color, d_radii);
        bad.struct_size = sizeof(bad) - 8;
        if (wg_rasterize_forward_ex(&bad) != WG_ERR_INVALID_ARGUMENT || wg_rasterize_forward_ex(nullptr) != WG_ERR_INVALID_ARGUMENT) {
            std::fprintf(stderr, "struct_size check missing\n");
            return 9;
        }
        const wg_call_options fast = {0, 0, 1};
        wg_backward_args mism = bwd_args(D, M, R, d_shs, nullptr, d_scales, d_rots, geom.p, bin.p, img.p, gcon, gcol, gsh);
        mism.options = &fast;   // the frame above was composited with exact_compositing = 1
        if (wg_rasterize_backward_ex(&mism) != WG_ERR_INVALID_ARGUMENT) { std::fprintf(stderr, "forward / backward option mismatch not detected\n"); return 9; }
        if (wg_set_option("exact_compositing", 0) == WG_OK || wg_get_option("deterministic_backward") != -1) {
            std::fprintf(stderr, "a result-affecting switch is still process-wide\n");
            return 9;
        }
    }
    {   // deterministic backward (per-call option, backward only): three passes over the one frame, a synchronisation between them (what
        // retain_graph=True does), must agree bit for bit -- and with the atomic sums' gradients up to the order of the additions.
        // (Round 5: with the scratch taken from the device's DEFAULT memory pool the second pass lost sums -- api.hip: det_scratch_alloc;
        //  the concurrent callers further down then caught the library's own keep-everything pool doing the same: the scratch is hipMalloc blocks now.)
        auto fetch = [&](const float* d, size_t n, std::vector<float>& h) { h.resize(n); return hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost); };
        struct { const char* name; const float* d; size_t n; } arr[] = {{"dL_dmean2D", g2d, 3 * (size_t)P}, {"dL_dconic", gcon, 4 * (size_t)P},
            {"dL_dopacity", gop, (size_t)P}, {"dL_dcolor", gcol, 3 * (size_t)P}, {"dL_dmean3D", g3d, 3 * (size_t)P}, {"dL_dcov3D", gcov, 6 * (size_t)P},
            {"dL_dsh", gsh, 3 * (size_t)P * M}, {"dL_dscale", gsc, 3 * (size_t)P}, {"dL_drot", grot, 4 * (size_t)P}};
        constexpr int NA = 9;
        std::vector<float> ref[NA], first[NA], cur;
        CHECK_HIP(hipStreamSynchronize(stream));
        for (int a = 0; a < NA; a++) CHECK_HIP(fetch(arr[a].d, arr[a].n, ref[a]));
        const wg_call_options det = {1, 1, 1};
        for (int rep = 0; rep < 3; rep++) {
            wg_backward_args da = bwd_args(D, M, R, d_shs, nullptr, d_scales, d_rots, geom.p, bin.p, img.p, gcon, gcol, gsh);
            da.options = &det;
            const int sd = wg_rasterize_backward_ex(&da);
            if (sd != WG_OK) { std::fprintf(stderr, "deterministic backward: %s (%s)\n", wg_status_string(sd), wg_last_hip_error()); return 16; }
            CHECK_HIP(hipStreamSynchronize(stream));
            for (int a = 0; a < NA; a++) {
                CHECK_HIP(fetch(arr[a].d, arr[a].n, rep ? cur : first[a]));
                if (!rep) continue;
                size_t bad = 0, k0 = 0;
                for (size_t k = 0; k < arr[a].n; k++)
                    if (std::memcmp(&first[a][k], &cur[k], 4)) { if (!bad++) k0 = k; }
                if (bad) {
                    std::fprintf(stderr, "deterministic backward: pass %d differs from pass 0 in %zu of %zu elements of %s, first at %zu: %.9g vs %.9g\n", rep, bad,
                                 arr[a].n, arr[a].name, k0, cur[k0], first[a][k0]);
                    return 16;
                }
            }
        }
        for (int a = 0; a < NA; a++) {
            double mx = 0, md = 0;
            for (size_t k = 0; k < arr[a].n; k++) { mx = std::fmax(mx, std::fabs(ref[a][k])); md = std::fmax(md, std::fabs(first[a][k] - ref[a][k])); }
            if (!(md <= 1e-4 * mx)) { std::fprintf(stderr, "deterministic backward: %s off the atomic sums by %g of %g\n", arr[a].name, md, mx); return 16; }
        }
        if (wg_set_option("release_scratch", 1) != WG_OK) { std::fprintf(stderr, "release_scratch: %s\n", wg_last_hip_error()); return 16; }
        // and an atomic-sum call straight after it
        wg_backward_args pa = bwd_args(D, M, R, d_shs, nullptr, d_scales, d_rots, geom.p, bin.p, img.p, gcon, gcol, gsh);
        if (wg_rasterize_backward_ex(&pa) != WG_OK) { std::fprintf(stderr, "plain backward after the deterministic ones: %s\n", wg_last_hip_error()); return 16; }
    }
    // geometry reuse: the same Gaussians and camera with other (precomputed) colours ride on the first call's projection and binning
    {
        std::vector<float> cols(3 * (size_t)P);
        for (auto& v : cols) v = uni(seed);
        float *d_cols, *d_color2;
        if (upload(cols, &d_cols)) return 2;
        CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&d_color2), (size_t)3 * W * H * sizeof(float)));
        Grow geom2;
        wg_forward_args ra = fwd_args(&geom2, nullptr, nullptr, 0, 0, nullptr, d_cols, nullptr, nullptr, nullptr, d_color2, nullptr);
        const wg_recolor_parent parent = {geom.p, bin.p, img.p, R};
        ra.recolor = &parent;
        const int R2 = wg_rasterize_forward_ex(&ra);
        if (R2 != R) { std::fprintf(stderr, "recolor: %s (%s)\n", wg_status_string(R2), wg_last_hip_error()); return 10; }
        const int st2 = wg_rasterize_backward(P, 0, 0, R, d_bg, W, H, d_means, nullptr, d_cols, d_scales, 1.0f, d_rots, nullptr, d_view, d_proj, d_campos, tanx,
                                              tany, 0.1f, nullptr, d_radii, geom2.p, bin.p, img.p, d_cot, g2d, gcon, gop, gcol, g3d, gcov, nullptr, gsc, grot,
                                              0, stream);
        if (st2 != WG_OK) { std::fprintf(stderr, "backward after recolor: %s (%s)\n", wg_status_string(st2), wg_last_hip_error()); return 11; }
        CHECK_HIP(hipStreamSynchronize(stream));
        std::vector<float> c2((size_t)3 * W * H);
        CHECK_HIP(hipMemcpy(c2.data(), d_color2, c2.size() * 4, hipMemcpyDeviceToHost));
        double s2 = 0;
        for (float v : c2) { if (!std::isfinite(v)) { std::fprintf(stderr, "non-finite recoloured image\n"); return 12; } s2 += v; }
        if (!(s2 > 0)) { std::fprintf(stderr, "empty recoloured image\n"); return 12; }
        (void)hipFree(d_cols); (void)hipFree(d_color2); (void)hipFree(geom2.p);
    }
    // round 4: two colour sets in ONE call, and get_gaussians() inside the preprocess kernels.  Self-consistency, bit for bit: each image of
    // the two-colour call equals the plain call's with that colour set; the raw-parameter call equals activations + plain call.
    {
        std::vector<float> c1(3 * (size_t)P), c2(3 * (size_t)P), filt(P), lsc(3 * (size_t)P), lop(P), rrot(4 * (size_t)P);
        for (auto& v : c1) v = uni(seed);
        for (auto& v : c2) v = uni(seed);
        for (int i = 0; i < P; i++) {
            filt[i] = 0.3f * scales[3 * i] * uni(seed);
            for (int k = 0; k < 3; k++) lsc[3 * i + k] = std::log(scales[3 * i + k]);
            lop[i] = std::log(opac[i] / (1.0f - opac[i]));
            const float m = 0.5f + uni(seed);
            for (int k = 0; k < 4; k++) rrot[4 * i + k] = m * rots[4 * i + k];
        }
        float *d_c1, *d_c2, *d_filt, *d_lsc, *d_lop, *d_rrot, *d_asc, *d_aop, *d_arot, *imgA, *imgB, *imgC, *imgD, *d_gc2;
        if (upload(c1, &d_c1) || upload(c2, &d_c2) || upload(filt, &d_filt) || upload(lsc, &d_lsc) || upload(lop, &d_lop) || upload(rrot, &d_rrot)) return 2;
        const size_t ibytes = (size_t)3 * W * H * sizeof(float);
        for (float** q : {&imgA, &imgB, &imgC, &imgD}) CHECK_HIP(hipMalloc(reinterpret_cast<void**>(q), ibytes));
        CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&d_asc), (size_t)P * 12));
        CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&d_aop), (size_t)P * 4));
        CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&d_arot), (size_t)P * 16));
        CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&d_gc2), (size_t)P * 12));
        Grow g2, b2, i2;
        auto plain = [&](const float* cols, const float* op, const float* sc, const float* rt, float* out) {
            return wg_rasterize_forward(Grow::alloc, &g2, Grow::alloc, &b2, Grow::alloc, &i2, P, 0, 0, d_bg, W, H, d_means, nullptr, cols, op, sc, 1.0f, rt,
                                        nullptr, d_view, d_proj, d_campos, tanx, tany, 0.1f, nullptr, 0, out, nullptr, 0, stream);
        };
        auto same = [&](const float* a, const float* b, const char* what) {
            std::vector<float> ha((size_t)3 * W * H), hb(ha.size());
            if (hipStreamSynchronize(stream) != hipSuccess || hipMemcpy(ha.data(), a, ibytes, hipMemcpyDeviceToHost) != hipSuccess ||
                hipMemcpy(hb.data(), b, ibytes, hipMemcpyDeviceToHost) != hipSuccess) return false;
            double sum = 0;
            for (size_t k = 0; k < ha.size(); k++) {
                if (ha[k] != hb[k]) { std::fprintf(stderr, "%s: images differ at %zu (%g vs %g)\n", what, k, ha[k], hb[k]); return false; }
                sum += ha[k];
            }
            if (!(sum > 0)) { std::fprintf(stderr, "%s: empty image\n", what); return false; }
            return true;
        };
        if (plain(d_c1, d_opac, d_scales, d_rots, imgA) <= 0 || plain(d_c2, d_opac, d_scales, d_rots, imgB) <= 0) return 13;
        wg_second_image sec = {d_c2, imgD, nullptr, nullptr};
        wg_forward_args da = fwd_args(&g2, &b2, &i2, 0, 0, nullptr, d_c1, d_opac, d_scales, d_rots, imgC, d_radii);
        da.second = &sec;
        const int Rd = wg_rasterize_forward_ex(&da);
        if (Rd != R) { std::fprintf(stderr, "two-colour forward: %d (%s)\n", Rd, wg_last_hip_error()); return 13; }
        if (!same(imgA, imgC, "two-colour call, first set") || !same(imgB, imgD, "two-colour call, second set")) return 13;
        sec.dL_dpix2 = d_cot; sec.dL_dcolor2 = d_gc2;
        wg_backward_args db = bwd_args(0, 0, Rd, nullptr, d_c1, d_scales, d_rots, g2.p, b2.p, i2.p, nullptr, gcol, nullptr);
        db.second = &sec;
        if (wg_rasterize_backward_ex(&db) != WG_OK) {
            std::fprintf(stderr, "two-colour backward: %s\n", wg_last_hip_error());
            return 14;
        }
        {
            std::vector<float> h(3 * (size_t)P);
            CHECK_HIP(hipStreamSynchronize(stream));
            CHECK_HIP(hipMemcpy(h.data(), d_gc2, h.size() * 4, hipMemcpyDeviceToHost));
            double l1 = 0;
            for (float v : h) { if (!std::isfinite(v)) { std::fprintf(stderr, "non-finite dL_dcolor2\n"); return 14; } l1 += std::fabs(v); }
            if (!(l1 > 0)) { std::fprintf(stderr, "dL_dcolor2 is zero\n"); return 14; }
        }
        // raw parameters: activations + 3-D filter by the stand-alone kernel, then the plain call -- against the raw-parameter call
        if (wg_activations_forward(P, d_rrot, d_lsc, d_lop, d_filt, d_arot, d_asc, d_aop, stream) != WG_OK) return 15;
        if (plain(d_c1, d_aop, d_asc, d_arot, imgA) <= 0) return 15;
        wg_raw_gaussians rawg = {d_filt, d_lop};
        wg_forward_args rfa = fwd_args(&g2, &b2, &i2, 0, 0, nullptr, d_c1, d_lop, d_lsc, d_rrot, imgC, d_radii);
        rfa.raw = &rawg;
        const int Rr = wg_rasterize_forward_ex(&rfa);
        if (Rr <= 0) { std::fprintf(stderr, "raw-parameter forward: %d (%s)\n", Rr, wg_last_hip_error()); return 15; }
        if (!same(imgA, imgC, "raw-parameter call")) return 15;
        wg_backward_args rba = bwd_args(0, 0, Rr, nullptr, d_c1, d_lsc, d_rrot, g2.p, b2.p, i2.p, nullptr, gcol, nullptr);
        rba.raw = &rawg;
        if (wg_rasterize_backward_ex(&rba) != WG_OK) {
            std::fprintf(stderr, "raw-parameter backward: %s\n", wg_last_hip_error());
            return 16;
        }
        // two tones of one SH block in ONE call: each image equals the toned call's with that tone, bit for bit
        {
            std::vector<float> mul(3 * (size_t)P), off(3 * (size_t)P);
            for (auto& v : mul) v = 0.5f + uni(seed);
            for (auto& v : off) v = 0.4f * (uni(seed) - 0.5f);
            float *d_mul, *d_off, *d_gmul, *d_goff;
            if (upload(mul, &d_mul) || upload(off, &d_off)) return 2;
            CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&d_gmul), (size_t)P * 12));
            CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&d_goff), (size_t)P * 12));
            wg_sh_tone t1 = {d_mul, d_off, 1.0f, 1.0f, d_gmul, d_goff};
            wg_sh_tone t2 = {nullptr, nullptr, 0.2f, INFINITY, nullptr, nullptr};
            auto toned = [&](const wg_sh_tone* t, float* out) {
                wg_forward_args ta = fwd_args(&g2, &b2, &i2, D, M, d_shs, nullptr, d_opac, d_scales, d_rots, out, nullptr);
                ta.tone = t;
                return wg_rasterize_forward_ex(&ta);
            };
            if (toned(&t1, imgA) != R || toned(&t2, imgB) != R) return 17;
            wg_second_image sec2 = {nullptr, imgD, d_cot, nullptr};
            wg_forward_args tta = fwd_args(&g2, &b2, &i2, D, M, d_shs, nullptr, d_opac, d_scales, d_rots, imgC, d_radii);
            tta.tone = &t1; tta.tone2 = &t2; tta.sh_second = 1; tta.second = &sec2;
            const int Rt = wg_rasterize_forward_ex(&tta);
            if (Rt != R) { std::fprintf(stderr, "two-tone forward: %d (%s)\n", Rt, wg_last_hip_error()); return 17; }
            if (!same(imgA, imgC, "two-tone call, first tone") || !same(imgB, imgD, "two-tone call, second tone")) return 17;
            wg_backward_args ttb = bwd_args(D, M, Rt, d_shs, nullptr, d_scales, d_rots, g2.p, b2.p, i2.p, nullptr, nullptr, gsh);
            ttb.tone = &t1; ttb.tone2 = &t2; ttb.sh_second = 1; ttb.second = &sec2;
            if (wg_rasterize_backward_ex(&ttb) != WG_OK) {
                std::fprintf(stderr, "two-tone backward: %s\n", wg_last_hip_error());
                return 18;
            }
            std::vector<float> h(3 * (size_t)P);
            CHECK_HIP(hipStreamSynchronize(stream));
            CHECK_HIP(hipMemcpy(h.data(), d_gmul, h.size() * 4, hipMemcpyDeviceToHost));
            double l1 = 0;
            for (float v : h) { if (!std::isfinite(v)) { std::fprintf(stderr, "non-finite dL_dmul\n"); return 18; } l1 += std::fabs(v); }
            if (!(l1 > 0)) { std::fprintf(stderr, "dL_dmul is zero\n"); return 18; }
            for (void* q : {(void*)d_mul, (void*)d_off, (void*)d_gmul, (void*)d_goff}) (void)hipFree(q);
        }
        // (d_radii was overwritten by these calls: restore the SH frame's for the dump below)
        if (wg_rasterize_forward(Grow::alloc, &geom, Grow::alloc, &bin, Grow::alloc, &img, P, D, M, d_bg, W, H, d_means, d_shs, nullptr, d_opac, d_scales, 1.0f,
                                 d_rots, nullptr, d_view, d_proj, d_campos, tanx, tany, 0.1f, nullptr, 0, d_color, d_radii, 0, stream) != R) return 4;
        CHECK_HIP(hipStreamSynchronize(stream));
        for (void* q : {(void*)d_c1, (void*)d_c2, (void*)d_filt, (void*)d_lsc, (void*)d_lop, (void*)d_rrot, (void*)d_asc, (void*)d_aop, (void*)d_arot, (void*)imgA,
                        (void*)imgB, (void*)imgC, (void*)imgD, (void*)d_gc2, (void*)g2.p, (void*)b2.p, (void*)i2.p})
            (void)hipFree(q);
    }
    // (the backward pass after the recolouring overwrote the gradient buffers: run the SH call's backward again for the dump below)
    if (wg_rasterize_backward(P, D, M, R, d_bg, W, H, d_means, d_shs, nullptr, d_scales, 1.0f, d_rots, nullptr, d_view, d_proj, d_campos, tanx, tany, 0.1f,
                              nullptr, d_radii, geom.p, bin.p, img.p, d_cot, g2d, gcon, gop, gcol, g3d, gcov, gsh, gsc, grot, 0, stream) != WG_OK)
        return 5;
    unsigned char* d_vis;
    CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&d_vis), (size_t)P));
    if (wg_mark_visible(P, d_means, d_view, d_proj, d_vis, stream) != WG_OK) return 6;
    CHECK_HIP(hipStreamSynchronize(stream));

    std::vector<float> color((size_t)3 * W * H), gm(3 * (size_t)P);
    std::vector<int> radii(P);
    std::vector<unsigned char> vis(P);
    CHECK_HIP(hipMemcpy(color.data(), d_color, color.size() * 4, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(gm.data(), g3d, gm.size() * 4, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(radii.data(), d_radii, radii.size() * 4, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(vis.data(), d_vis, vis.size(), hipMemcpyDeviceToHost));
    double sum = 0, gsum = 0;
    int nvis = 0, nrad = 0;
    for (float v : color) { if (!std::isfinite(v)) { std::fprintf(stderr, "non-finite colour\n"); return 7; } sum += v; }
    for (float v : gm) { if (!std::isfinite(v)) { std::fprintf(stderr, "non-finite gradient\n"); return 7; } gsum += std::fabs(v); }
    for (int i = 0; i < P; i++) { nvis += vis[i]; nrad += radii[i] > 0; }
    if (nrad == 0 || nvis < nrad || gsum <= 0) { std::fprintf(stderr, "implausible outputs: nrad %d nvis %d gsum %g\n", nrad, nvis, gsum); return 8; }
    {   // concurrent callers: three host threads, a stream and scratch buffers each, the same frame, eight forward + backward calls each, all
        // in flight together (no interpreter lock in the way): every image bit for bit the one above, dL_dmean3D within the atomic sums'
        // rounding -- thread 1 in the deterministic mode (bit for bit across its calls).
        // Diagnostic knobs (environment; defaults = the test): WG_DRV_THREADS (3), WG_DRV_ITERS (8), WG_DRV_DET_MASK (bit t: caller t runs the
        // deterministic backward; 2), WG_DRV_REPEAT (the whole block that many times; 1), WG_DRV_KEEP (1: callers free nothing until all have
        // joined), WG_DRV_SHARED_STREAM (1: all callers feed ONE stream), WG_DRV_VERBOSE (1: every deviating call is described -- how many elements, where, what was there -- and the run goes on).
        auto env_int = [](const char* n, int d) { const char* v = std::getenv(n); return v && *v ? std::atoi(v) : d; };
        const int n_threads = env_int("WG_DRV_THREADS", 3), n_iters = env_int("WG_DRV_ITERS", 8), det_mask = env_int("WG_DRV_DET_MASK", 2),
                  n_repeat = env_int("WG_DRV_REPEAT", 1), keep = env_int("WG_DRV_KEEP", 0), verbose = env_int("WG_DRV_VERBOSE", 0),
                  shared_stream = env_int("WG_DRV_SHARED_STREAM", 0);
        hipStream_t one_stream = nullptr;
        if (shared_stream) CHECK_HIP(hipStreamCreate(&one_stream));
        std::atomic<int> failures{0}, bad_calls{0};
        std::vector<void*> kept;
        std::mutex kept_mu;
        auto caller = [&](int tid) {
            hipStream_t st;
            Grow g, b, i;
            float *out = nullptr, *t2d = nullptr, *tcon = nullptr, *top = nullptr, *tcol = nullptr, *t3d = nullptr, *tcov = nullptr, *tsh = nullptr, *tsc = nullptr, *trot = nullptr;
            int* rad = nullptr;
            bool ok = (shared_stream ? (st = one_stream, true) : hipStreamCreate(&st) == hipSuccess) && hipMalloc(reinterpret_cast<void**>(&out), (size_t)3 * W * H * 4) == hipSuccess &&
                      hipMalloc(reinterpret_cast<void**>(&rad), (size_t)P * 4) == hipSuccess;
            const std::pair<float**, size_t> gr[] = {{&t2d, 3u * (size_t)P}, {&tcon, 4u * (size_t)P}, {&top, (size_t)P}, {&tcol, 3u * (size_t)P}, {&t3d, 3u * (size_t)P},
                                                     {&tcov, 6u * (size_t)P}, {&tsh, (size_t)P * M * 3}, {&tsc, 3u * (size_t)P}, {&trot, 4u * (size_t)P}};
            for (const auto& q : gr) ok = ok && hipMalloc(reinterpret_cast<void**>(q.first), q.second * 4) == hipSuccess;
            std::vector<float> himg(color.size()), hg(gm.size()), hg0;
            const wg_call_options det = {1, 1, 1};
            const bool is_det = ((det_mask >> tid) & 1) != 0;
            for (int it = 0; ok && it < n_iters; it++) {
                wg_forward_args fa{};
                fa.struct_size = sizeof(fa);
                fa.geometry_alloc = Grow::alloc; fa.geometry_user = &g; fa.binning_alloc = Grow::alloc; fa.binning_user = &b; fa.image_alloc = Grow::alloc; fa.image_user = &i;
                fa.P = P; fa.D = D; fa.M = M; fa.width = W; fa.height = H; fa.scale_modifier = 1.0f; fa.tan_fovx = tanx; fa.tan_fovy = tany; fa.kernel_size = 0.1f;
                fa.background = d_bg; fa.means3D = d_means; fa.shs = d_shs; fa.opacities = d_opac; fa.scales = d_scales; fa.rotations = d_rots;
                fa.viewmatrix = d_view; fa.projmatrix = d_proj; fa.cam_pos = d_campos; fa.out_color = out; fa.radii = rad; fa.stream = st;
                const int Rt = wg_rasterize_forward_ex(&fa);
                wg_backward_args ba{};
                ba.struct_size = sizeof(ba);
                ba.P = P; ba.D = D; ba.M = M; ba.R = Rt; ba.width = W; ba.height = H; ba.scale_modifier = 1.0f; ba.tan_fovx = tanx; ba.tan_fovy = tany; ba.kernel_size = 0.1f;
                ba.background = d_bg; ba.means3D = d_means; ba.shs = d_shs; ba.scales = d_scales; ba.rotations = d_rots; ba.viewmatrix = d_view; ba.projmatrix = d_proj;
                ba.campos = d_campos; ba.radii = rad; ba.geom_buffer = g.p; ba.binning_buffer = b.p; ba.image_buffer = i.p; ba.dL_dpix = d_cot;
                ba.dL_dmean2D = t2d; ba.dL_dconic = tcon; ba.dL_dopacity = top; ba.dL_dcolor = tcol; ba.dL_dmean3D = t3d; ba.dL_dcov3D = tcov; ba.dL_dsh = tsh;
                ba.dL_dscale = tsc; ba.dL_drot = trot; ba.stream = st;
                if (is_det) ba.options = &det;
                ok = Rt == R && wg_rasterize_backward_ex(&ba) == WG_OK && hipStreamSynchronize(st) == hipSuccess &&
                     hipMemcpy(himg.data(), out, himg.size() * 4, hipMemcpyDeviceToHost) == hipSuccess &&
                     hipMemcpy(hg.data(), t3d, hg.size() * 4, hipMemcpyDeviceToHost) == hipSuccess;
                if (!ok) { std::fprintf(stderr, "concurrent caller %d, call %d: R %d (%s)\n", tid, it, Rt, wg_last_hip_error()); break; }
                bool call_ok = true;
                if (std::memcmp(himg.data(), color.data(), himg.size() * 4)) { std::fprintf(stderr, "concurrent caller %d, call %d: another image\n", tid, it); call_ok = false; }
                double mx = 0, md = 0;
                for (size_t k = 0; k < hg.size(); k++) { mx = std::fmax(mx, std::fabs(gm[k])); md = std::fmax(md, std::fabs(hg[k] - gm[k])); }
                if (!(md <= 1e-5 * mx)) {
                    std::fprintf(stderr, "concurrent caller %d%s, call %d: dL_dmean3D off by %g of %g\n", tid, is_det ? " (deterministic)" : "", it, md, mx);
                    call_ok = false;
                    if (verbose) {   // how many Gaussians, where, and what stood there
                        size_t n_off = 0, n_zero = 0, first = hg.size(), last = 0;
                        for (size_t k = 0; k < hg.size(); k++)
                            if (std::fabs(hg[k] - gm[k]) > 1e-5 * mx) { n_off++; n_zero += hg[k] == 0.0f; first = std::min(first, k / 3); last = std::max(last, k / 3); }
                        std::fprintf(stderr, "    %zu elements off (%zu of them zero) in Gaussians %zu .. %zu of %d;", n_off, n_zero, first, last, P);
                        int shown = 0;
                        for (size_t k = 0; k < hg.size() && shown < 6; k++)
                            if (std::fabs(hg[k] - gm[k]) > 1e-5 * mx) { std::fprintf(stderr, " [%zu] %g vs %g", k, hg[k], gm[k]); shown++; }
                        std::fprintf(stderr, "\n");
                    }
                }
                if (is_det) {
                    if (hg0.empty()) { if (call_ok) hg0 = hg; }
                    else if (std::memcmp(hg0.data(), hg.data(), hg.size() * 4)) { std::fprintf(stderr, "concurrent caller %d, call %d: deterministic gradients differ\n", tid, it); call_ok = false; }
                }
                if (!call_ok) { bad_calls++; if (!verbose) { ok = false; break; } }
            }
            if (!ok) failures++;
            std::vector<void*> mine = {(void*)out, (void*)rad, (void*)g.p, (void*)b.p, (void*)i.p};
            for (const auto& q : gr) mine.push_back(*q.first);
            if (keep) { std::lock_guard<std::mutex> l(kept_mu); kept.insert(kept.end(), mine.begin(), mine.end()); }
            else for (void* q : mine) (void)hipFree(q);
            if (!shared_stream) (void)hipStreamDestroy(st);
        };
        for (int rep = 0; rep < n_repeat; rep++) {
            std::vector<std::thread> th;
            for (int t = 0; t < n_threads; t++) th.emplace_back(caller, t);
            for (auto& t : th) t.join();
        }
        for (void* q : kept) (void)hipFree(q);
        if (shared_stream) (void)hipStreamDestroy(one_stream);
        if (verbose) std::fprintf(stderr, "concurrent block: %d deviating call(s) in %d x %d x %d\n", bad_calls.load(), n_repeat, n_threads, n_iters);
        if (failures.load() || bad_calls.load()) { std::fprintf(stderr, "%d of %d concurrent callers failed\n", failures.load(), n_threads); return 17; }
    }
    if (argc > 4) {
        std::FILE* f = std::fopen(argv[4], "wb");
        if (!f) { std::fprintf(stderr, "cannot open %s\n", argv[4]); return 9; }
        const int32_t hdr[6] = {P, W, H, D, M, R};
        const float fov[2] = {tanx, tany};
        std::fwrite(hdr, sizeof(hdr), 1, f);
        std::fwrite(fov, sizeof(fov), 1, f);
        for (const std::vector<float>* v : {&means, &scales, &rots, &opac, &shs, &view, &proj, &campos, &bg, &cot}) std::fwrite(v->data(), 4, v->size(), f);
        std::fwrite(color.data(), 4, color.size(), f);
        const std::pair<float*, size_t> outs[] = {{g2d, 3u * (size_t)P}, {gcon, 4u * (size_t)P}, {gop, (size_t)P}, {gcol, 3u * (size_t)P}, {g3d, 3u * (size_t)P},
                                                  {gcov, 6u * (size_t)P}, {gsh, (size_t)P * M * 3}, {gsc, 3u * (size_t)P}, {grot, 4u * (size_t)P}};
        for (const auto& o : outs) {
            std::vector<float> h(o.second);
            CHECK_HIP(hipMemcpy(h.data(), o.first, o.second * 4, hipMemcpyDeviceToHost));
            std::fwrite(h.data(), 4, h.size(), f);
        }
        std::fwrite(radii.data(), 4, radii.size(), f);
        std::fwrite(vis.data(), 1, vis.size(), f);
        std::fclose(f);
    }
    std::printf("ok num_rendered=%d visible=%d radii>0=%d checksum=%.6f grad_l1=%.6e\n", R, nvis, nrad, sum, gsum);
    for (void* p : {(void*)d_means, (void*)d_scales, (void*)d_rots, (void*)d_opac, (void*)d_shs, (void*)d_view, (void*)d_proj, (void*)d_campos, (void*)d_bg,
                    (void*)d_cot, (void*)d_color, (void*)d_radii, (void*)g2d, (void*)gcon, (void*)gop, (void*)gcol, (void*)g3d, (void*)gcov, (void*)gsh,
                    (void*)gsc, (void*)grot, (void*)d_vis, (void*)geom.p, (void*)bin.p, (void*)img.p})
        (void)hipFree(p);
    (void)hipStreamDestroy(stream);
    return 0;
}
