// oracle/ref_hip/driver.cpp -- TEST INFRASTRUCTURE ONLY (never linked into, imported by or shipped with the product).
//
// A torch-free C-ABI around the REAL reference rasterizer: CudaRasterizer::Rasterizer::{forward,backward,markVisible}
// (/root/reference/submodules/diff-gaussian-rasterization/cuda_rasterizer/rasterizer.h:18-90), whose three CUDA sources
// oracle/ref_hip/Makefile compiles for gfx950 IN PLACE with hipcc (shim/ gives the CUDA header names a HIP body).  It plays
// the part of the reference's torch binding (rasterize_points.cu:35-117 forward, :119-204 backward, :206-224 markVisible):
// grow-only scratch buffers behind the three std::function allocators, zero-initialised gradient outputs, the same argument
// order.  Uses: (1) scripts/make_golden_from_reference.py runs it on the GPU box and writes tests/golden/ref_hip_*.npz, the
// fixtures that PIN oracle/wg_oracle.c against outputs of the reference itself; (2) scripts/bench_reference_hip.py times the
// reference's own kernels on the MI355X beside ours.  All pointers are device pointers; everything runs on the null stream,
// as the reference does.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <functional>
#include "rasterizer.h"

namespace {

struct Buffer {
    char* ptr = nullptr;
    size_t cap = 0;
    char* resize(size_t n) {  // torch's resize_(): grow-only, contents not preserved
        if (n > cap) {
            if (ptr) hipFree(ptr);
            ptr = nullptr;
            cap = 0;
            if (hipMalloc(&ptr, n + 256) != hipSuccess) return nullptr;
            cap = n;
        }
        return ptr;
    }
};

Buffer g_geom, g_binning, g_image;
int g_rendered = 0;

template <typename T>
T* align128(char* p) {
    return reinterpret_cast<T*>((reinterpret_cast<uintptr_t>(p) + 127) & ~(uintptr_t)127);
}

}  // namespace

extern "C" {

// Returns num_rendered (< 0: allocation failure).  out_final_T / out_n_contrib (N floats / N uint32, may be null) receive the
// first two arrays of the image buffer (ImageState::fromChunk, rasterizer_impl.cu:172-179): what the Python layer reads the
// accumulation from (diff_gaussian_rasterization/__init__.py:101-112).
int refhip_forward(int P, int D, int M, const float* background, int W, int H, const float* means3D, const float* shs,
                   const float* colors_precomp, const float* opacities, const float* scales, float scale_modifier,
                   const float* rotations, const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                   const float* cam_pos, float tan_fovx, float tan_fovy, float kernel_size, const float* subpixel_offset,
                   int prefiltered, float* out_color, int* radii, float* out_final_T, uint32_t* out_n_contrib, int debug) {
    const size_t N = (size_t)W * H;
    hipMemsetAsync(out_color, 0, 3 * N * sizeof(float), 0);  // torch::full(0) of the binding
    hipMemsetAsync(radii, 0, (size_t)P * sizeof(int), 0);
    g_rendered = 0;
    if (P == 0) return 0;
    bool failed = false;
    auto alloc = [&failed](Buffer& b) {
        return std::function<char*(size_t)>([&b, &failed](size_t n) {
            char* p = b.resize(n);
            if (!p) failed = true;
            return p;
        });
    };
    g_rendered = CudaRasterizer::Rasterizer::forward(alloc(g_geom), alloc(g_binning), alloc(g_image), P, D, M, background, W, H,
                                                     means3D, shs, colors_precomp, opacities, scales, scale_modifier, rotations,
                                                     cov3D_precomp, viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy,
                                                     kernel_size, subpixel_offset, prefiltered != 0, out_color, radii,
                                                     debug != 0);
    if (failed) return -1;
    if (out_final_T) {
        float* acc = align128<float>(g_image.ptr);
        hipMemcpyAsync(out_final_T, acc, N * sizeof(float), hipMemcpyDeviceToDevice, 0);
        if (out_n_contrib) {
            uint32_t* nc = align128<uint32_t>(reinterpret_cast<char*>(acc + N));
            hipMemcpyAsync(out_n_contrib, nc, N * sizeof(uint32_t), hipMemcpyDeviceToDevice, 0);
        }
    }
    return g_rendered;
}

// Backward of the LAST refhip_forward (its three buffers are still in place).  Gradient outputs are zeroed here, as the
// binding's torch::zeros do.  dL_dmean2D: P*3 (x, y, GOF |.| accumulation), dL_dconic: P*4, dL_dcov3D: P*6, dL_dsh: P*M*3.
void refhip_backward(int P, int D, int M, const float* background, int W, int H, const float* means3D, const float* shs,
                     const float* colors_precomp, const float* scales, float scale_modifier, const float* rotations,
                     const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                     float tan_fovx, float tan_fovy, float kernel_size, const float* subpixel_offset, const int* radii,
                     const float* dL_dpix, float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor,
                     float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot, int debug) {
    const size_t p = (size_t)P;
    hipMemsetAsync(dL_dmean3D, 0, p * 3 * sizeof(float), 0);
    hipMemsetAsync(dL_dmean2D, 0, p * 3 * sizeof(float), 0);
    hipMemsetAsync(dL_dcolor, 0, p * 3 * sizeof(float), 0);
    hipMemsetAsync(dL_dconic, 0, p * 4 * sizeof(float), 0);
    hipMemsetAsync(dL_dopacity, 0, p * sizeof(float), 0);
    hipMemsetAsync(dL_dcov3D, 0, p * 6 * sizeof(float), 0);
    if (M > 0) hipMemsetAsync(dL_dsh, 0, p * M * 3 * sizeof(float), 0);
    hipMemsetAsync(dL_dscale, 0, p * 3 * sizeof(float), 0);
    hipMemsetAsync(dL_drot, 0, p * 4 * sizeof(float), 0);
    if (P == 0) return;
    CudaRasterizer::Rasterizer::backward(P, D, M, g_rendered, background, W, H, means3D, shs, colors_precomp, scales,
                                         scale_modifier, rotations, cov3D_precomp, viewmatrix, projmatrix, cam_pos, tan_fovx,
                                         tan_fovy, kernel_size, subpixel_offset, radii, g_geom.ptr, g_binning.ptr, g_image.ptr,
                                         dL_dpix, dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh,
                                         dL_dscale, dL_drot, debug != 0);
}

void refhip_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix, unsigned char* present) {
    hipMemsetAsync(present, 0, (size_t)P, 0);
    if (P == 0) return;
    CudaRasterizer::Rasterizer::markVisible(P, const_cast<float*>(means3D), const_cast<float*>(viewmatrix),
                                            const_cast<float*>(projmatrix), reinterpret_cast<bool*>(present));
}

void refhip_release(void) {
    for (Buffer* b : {&g_geom, &g_binning, &g_image}) {
        if (b->ptr) hipFree(b->ptr);
        b->ptr = nullptr;
        b->cap = 0;
    }
}

}  // extern "C"
