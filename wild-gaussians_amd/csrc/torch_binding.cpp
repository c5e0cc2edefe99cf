// diff_gaussian_rasterization._C_torch -- a compiled torch binding of the C-ABI (include/wg_rasterizer.h), standing where the
// reference's pybind11 module stands (submodules/diff-gaussian-rasterization/ext.cpp:15-19, rasterize_points.{h,cu}): the same three
// functions, the same argument order and return tuples (rasterize_points.h:18-71).  It is INTEGRATION.md section 2 as a file that
// builds and is tested: what a maintainer of the reference would write to keep `rasterize_points.cu`'s surface on top of
// libwg_rasterizer.so.  Host-side C++ only (no kernels here): torch supplies device memory, the device guard and the current HIP stream.
// The reference's surface only -- the opt-ins beyond it (sh_tone, binning_capacity, colors2, filter_3D, geometry reuse) stay with the
// ctypes binding (_C.py), which also remains the default; WG_BINDING=torch selects this module for the plain calls.
#include <torch/extension.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>   // torch on ROCm calls its device type "cuda": the stream getter of that name
#include <c10/core/DeviceGuard.h>

#include <tuple>

#include "wg_rasterizer.h"

namespace {

// the role of resizeFunctional (rasterize_points.cu:27-33); the binning allocator may be called twice in one forward call
// (speculative forward): the tensor is simply resized again, the pointer returned last is the one in use
char* resize_cb(size_t n, void* user) {
    auto& t = *static_cast<torch::Tensor*>(user);
    t.resize_({static_cast<int64_t>(n)});
    return reinterpret_cast<char*>(t.data_ptr());
}

torch::Tensor f32(const torch::Tensor& t, const torch::Device& dev) {   // float32, contiguous, on dev; zero-sized = "absent"
    if (t.numel() == 0) return t;
    return t.to(dev, torch::kFloat32).contiguous();
}
const float* ptr(const torch::Tensor& t) { return t.numel() ? t.data_ptr<float>() : nullptr; }

void check(int status, const char* what) {
    if (status >= 0) return;
    std::string msg = std::string(what) + " failed: " + wg_status_string(status);
    if (status == -3) msg += std::string(": ") + wg_last_hip_error();
    throw std::runtime_error(msg);   // -> RuntimeError, as AT_ERROR / std::runtime_error in the reference
}

}  // namespace

// rasterize_points.cu:35-119
std::tuple<int, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor> RasterizeGaussiansHIP(
    const torch::Tensor& background, const torch::Tensor& means3D, const torch::Tensor& colors, const torch::Tensor& opacity,
    const torch::Tensor& scales, const torch::Tensor& rotations, const float scale_modifier, const torch::Tensor& cov3D_precomp,
    const torch::Tensor& viewmatrix, const torch::Tensor& projmatrix, const float tan_fovx, const float tan_fovy, const float kernel_size,
    const torch::Tensor& subpixel_offset, const int image_height, const int image_width, const torch::Tensor& sh, const int degree,
    const torch::Tensor& campos, const bool prefiltered, const bool debug) {
    if (means3D.ndimension() != 2 || means3D.size(1) != 3) AT_ERROR("means3D must have dimensions (num_points, 3)");   // :59-61
    if (!means3D.is_cuda()) throw std::runtime_error("means3D must live on a HIP device: this rasterizer has no CPU path");
    const auto dev = means3D.device();
    const c10::DeviceGuard guard(dev);
    const int P = static_cast<int>(means3D.size(0)), H = image_height, W = image_width;
    const auto bytes = torch::TensorOptions().dtype(torch::kUInt8).device(dev);
    torch::Tensor geomBuffer = torch::empty({0}, bytes), binningBuffer = torch::empty({0}, bytes), imgBuffer = torch::empty({0}, bytes);
    if (P == 0)   // :83: nothing is launched, the image stays zero
        return std::make_tuple(0, torch::zeros({3, H, W}, means3D.options().dtype(torch::kFloat32)),
                               torch::zeros({0}, means3D.options().dtype(torch::kInt32)), geomBuffer, binningBuffer, imgBuffer);
    // both outputs are fully written by the kernels: no zero fill
    torch::Tensor out_color = torch::empty({3, H, W}, means3D.options().dtype(torch::kFloat32));
    torch::Tensor radii = torch::empty({P}, means3D.options().dtype(torch::kInt32));
    const auto m3 = f32(means3D, dev), bg = f32(background, dev), col = f32(colors, dev), op = f32(opacity, dev), sc = f32(scales, dev),
               rot = f32(rotations, dev), cov = f32(cov3D_precomp, dev), vm = f32(viewmatrix, dev), pm = f32(projmatrix, dev),
               cam = f32(campos, dev), so = f32(subpixel_offset, dev), shs = f32(sh, dev);
    const int M = shs.numel() ? static_cast<int>(shs.size(1)) : 0;   // :85-89
    const int rendered = wg_rasterize_forward(
        resize_cb, &geomBuffer, resize_cb, &binningBuffer, resize_cb, &imgBuffer, P, degree, M, ptr(bg), W, H, ptr(m3), ptr(shs), ptr(col),
        ptr(op), ptr(sc), scale_modifier, ptr(rot), ptr(cov), ptr(vm), ptr(pm), ptr(cam), tan_fovx, tan_fovy, kernel_size, ptr(so),
        prefiltered ? 1 : 0, out_color.data_ptr<float>(), radii.data_ptr<int>(), debug ? 1 : 0, c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(dev.index()).stream());
    check(rendered, "wg_rasterize_forward");
    return std::make_tuple(rendered, out_color, radii, geomBuffer, binningBuffer, imgBuffer);
}

// rasterize_points.cu:121-204
std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor>
RasterizeGaussiansBackwardHIP(const torch::Tensor& background, const torch::Tensor& means3D, const torch::Tensor& radii,
                              const torch::Tensor& colors, const torch::Tensor& scales, const torch::Tensor& rotations, const float scale_modifier,
                              const torch::Tensor& cov3D_precomp, const torch::Tensor& viewmatrix, const torch::Tensor& projmatrix,
                              const float tan_fovx, const float tan_fovy, const float kernel_size, const torch::Tensor& subpixel_offset,
                              const torch::Tensor& dL_dout_color, const torch::Tensor& sh, const int degree, const torch::Tensor& campos,
                              const torch::Tensor& geomBuffer, const int R, const torch::Tensor& binningBuffer, const torch::Tensor& imageBuffer,
                              const bool debug) {
    const auto dev = means3D.device();
    const c10::DeviceGuard guard(dev);
    const int P = static_cast<int>(means3D.size(0));
    const int H = static_cast<int>(dL_dout_color.size(1)), W = static_cast<int>(dL_dout_color.size(2));
    const auto shs = f32(sh, dev);
    const int M = shs.numel() ? static_cast<int>(shs.size(1)) : 0;
    const auto f = means3D.options().dtype(torch::kFloat32);
    // the reference zero-fills nine tensors (:157-165); with the gradient record every one of them is fully written by the library
    // (zeros for culled Gaussians), without it the four accumulation targets must arrive zeroed
    const bool record = wg_get_option("grad_record") == 1 || wg_get_option("deterministic_backward") == 1;
    auto alloc = [&](std::initializer_list<int64_t> shape, bool accumulated) {
        return (P == 0 || (accumulated && !record)) ? torch::zeros(shape, f) : torch::empty(shape, f);
    };
    torch::Tensor dL_dmeans3D = alloc({P, 3}, false), dL_dmeans2D = alloc({P, 3}, true), dL_dcolors = alloc({P, 3}, true),
                  dL_dconic = record ? torch::Tensor() : torch::zeros({P, 2, 2}, f), dL_dopacity = alloc({P, 1}, true),
                  dL_dcov3D = alloc({P, 6}, false), dL_dsh = alloc({P, M, 3}, false);
    const bool have_scales = scales.numel() != 0;
    torch::Tensor dL_dscales = have_scales ? alloc({P, 3}, false) : torch::zeros({P, 3}, f),
                  dL_drotations = have_scales ? alloc({P, 4}, false) : torch::zeros({P, 4}, f);
    if (P != 0) {
        const auto m3 = f32(means3D, dev), bg = f32(background, dev), col = f32(colors, dev), sc = f32(scales, dev), rot = f32(rotations, dev),
                   cov = f32(cov3D_precomp, dev), vm = f32(viewmatrix, dev), pm = f32(projmatrix, dev), cam = f32(campos, dev),
                   so = f32(subpixel_offset, dev), dL = f32(dL_dout_color, dev);
        const auto rad = radii.contiguous();
        const int status = wg_rasterize_backward(
            P, degree, M, R, ptr(bg), W, H, ptr(m3), ptr(shs), ptr(col), ptr(sc), scale_modifier, ptr(rot), ptr(cov), ptr(vm), ptr(pm), ptr(cam),
            tan_fovx, tan_fovy, kernel_size, ptr(so), rad.numel() ? rad.data_ptr<int>() : nullptr,
            reinterpret_cast<char*>(geomBuffer.data_ptr()), reinterpret_cast<char*>(binningBuffer.data_ptr()),
            reinterpret_cast<char*>(imageBuffer.data_ptr()), ptr(dL), dL_dmeans2D.data_ptr<float>(),
            dL_dconic.defined() ? dL_dconic.data_ptr<float>() : nullptr, dL_dopacity.data_ptr<float>(), dL_dcolors.data_ptr<float>(),
            dL_dmeans3D.data_ptr<float>(), dL_dcov3D.data_ptr<float>(), M ? dL_dsh.data_ptr<float>() : nullptr, dL_dscales.data_ptr<float>(),
            dL_drotations.data_ptr<float>(), debug ? 1 : 0, c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(dev.index()).stream());
        check(status, "wg_rasterize_backward");
    }
    return std::make_tuple(dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations);   // :201
}

// rasterize_points.cu:206-225
torch::Tensor markVisible(torch::Tensor& means3D, torch::Tensor& viewmatrix, torch::Tensor& projmatrix) {
    const auto dev = means3D.device();
    const int P = static_cast<int>(means3D.size(0));
    torch::Tensor present = torch::zeros({P}, means3D.options().dtype(torch::kBool));
    if (P != 0) {
        const c10::DeviceGuard guard(dev);
        const auto m3 = f32(means3D, dev), vm = f32(viewmatrix, dev), pm = f32(projmatrix, dev);
        check(wg_mark_visible(P, ptr(m3), ptr(vm), ptr(pm), reinterpret_cast<unsigned char*>(present.data_ptr()),
                              c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(dev.index()).stream()), "wg_mark_visible");
    }
    return present;
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {   // ext.cpp:15-19
    m.def("rasterize_gaussians", &RasterizeGaussiansHIP);
    m.def("rasterize_gaussians_backward", &RasterizeGaussiansBackwardHIP);
    m.def("mark_visible", &markVisible);
}
