// diff_gaussian_rasterization._C_torch -- the compiled torch binding of the C-ABI (include/wg_rasterizer.h), standing where the
// reference's pybind11 module stands (submodules/diff-gaussian-rasterization/ext.cpp:15-19, rasterize_points.{h,cu}): the same three
// functions, the same argument order and return tuples (rasterize_points.h:18-71) -- INTEGRATION.md section 2 as a file that builds and
// is tested: what a maintainer of the reference would write to keep `rasterize_points.cu`'s surface on top of libwg_rasterizer.so --
// plus `rasterize_gaussians_ex` / `rasterize_gaussians_backward_ex`: the same two calls with everything beyond the reference by
// argument (the optional blocks and per-call options of wg_rasterize_forward_ex / _backward_ex).  It is the DEFAULT binding when built
// (diff_gaussian_rasterization/_C.py loads it; the ctypes code there is the fallback and keeps the Python-side geometry reuse).
// Host-side C++ only (no kernels here): torch supplies device memory, the device guard and the current HIP stream.
#include <torch/extension.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>   // torch on ROCm calls its device type "cuda": the stream getter of that name
#include <c10/core/DeviceGuard.h>

#include <limits>
#include <tuple>
#include <vector>

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
    const bool record = true;   // (the reference-shaped call runs with the default per-call options: gradient record on)
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


// ---- everything beyond the reference, by argument (the blocks of wg_forward_args / wg_backward_args) -------------------------------------
namespace {

namespace py = pybind11;
using OptTensor = c10::optional<torch::Tensor>;

// sh_tone = None | (mul [P,3] | None, offset [P,3] | None, pre_clamp_max | None, post_clamp_max | None)   (wg_sh_tone)
struct Tone {
    wg_sh_tone t{};
    torch::Tensor mul, offset, g_mul, g_offset;
    bool given = false;
    void parse(const py::object& o, const torch::Device& dev, int P, bool grads, const torch::TensorOptions& f) {
        if (o.is_none()) return;
        const py::tuple tu = o.cast<py::tuple>();
        if (tu.size() != 4) throw std::runtime_error("a tone is (mul, offset, pre_clamp_max, post_clamp_max)");
        given = true;
        const float inf = std::numeric_limits<float>::infinity();
        auto take = [&](int i, torch::Tensor& dst, const float*& p, const char* name) {
            if (tu[i].is_none()) return;
            dst = f32(tu[i].cast<torch::Tensor>(), dev);
            if (dst.numel() != 3 * (int64_t)P) throw std::runtime_error(std::string("sh_") + name + " must have 3 * P elements");
            p = dst.data_ptr<float>();
        };
        take(0, mul, t.mul, "mul");
        take(1, offset, t.offset, "offset");
        t.pre_clamp_max = tu[2].is_none() ? inf : tu[2].cast<float>();
        t.post_clamp_max = tu[3].is_none() ? inf : tu[3].cast<float>();
        if (grads) {
            auto alloc = [&](bool have) { return have ? (P ? torch::empty({P, 3}, f) : torch::zeros({P, 3}, f)) : torch::Tensor(); };
            g_mul = alloc(t.mul != nullptr);
            g_offset = alloc(t.offset != nullptr);
            t.dL_dmul = g_mul.defined() ? g_mul.data_ptr<float>() : nullptr;
            t.dL_doffset = g_offset.defined() ? g_offset.data_ptr<float>() : nullptr;
        }
    }
    py::object grad(const torch::Tensor& g) const { return g.defined() ? py::cast(g) : py::none(); }
};

wg_call_options call_options(const std::tuple<int, int, int>& o) {
    wg_call_options c;
    c.exact_compositing = std::get<0>(o); c.deterministic_backward = std::get<1>(o); c.grad_record = std::get<2>(o);
    return c;
}

}  // namespace

py::tuple RasterizeGaussiansEx(const torch::Tensor& background, const torch::Tensor& means3D, const torch::Tensor& colors, const torch::Tensor& opacity,
                               const torch::Tensor& scales, const torch::Tensor& rotations, const float scale_modifier, const torch::Tensor& cov3D_precomp,
                               const torch::Tensor& viewmatrix, const torch::Tensor& projmatrix, const float tan_fovx, const float tan_fovy,
                               const float kernel_size, const OptTensor& subpixel_offset, const int image_height, const int image_width,
                               const torch::Tensor& sh, const int degree, const torch::Tensor& campos, const bool prefiltered, const bool debug,
                               const py::object& sh_tone, const py::object& binning_capacity, const OptTensor& colors2, const OptTensor& filter_3D,
                               const py::object& sh_second, const std::tuple<int, int, int>& options) {
    if (means3D.ndimension() != 2 || means3D.size(1) != 3) AT_ERROR("means3D must have dimensions (num_points, 3)");   // rasterize_points.cu:59-61
    if (!means3D.is_cuda()) throw std::runtime_error("means3D must live on a HIP device: this rasterizer has no CPU path");
    const auto dev = means3D.device();
    const c10::DeviceGuard guard(dev);
    const int P = static_cast<int>(means3D.size(0)), H = image_height, W = image_width;
    const auto f = means3D.options().dtype(torch::kFloat32);
    const auto bytes = torch::TensorOptions().dtype(torch::kUInt8).device(dev);
    torch::Tensor geomBuffer = torch::empty({0}, bytes), binningBuffer = torch::empty({0}, bytes), imgBuffer = torch::empty({0}, bytes);
    const bool two = colors2.has_value() || !sh_second.is_none();
    // The checks of the ctypes binding (_C.py), same messages: a wrong-size tensor must raise, not send a kernel out of bounds.
    {
        const int64_t P64 = P;
        const bool fixed = !binning_capacity.is_none();
        if (filter_3D.has_value() && (colors2.has_value() || fixed || scales.numel() == 0 || filter_3D->numel() != P64))
            throw std::runtime_error("filter_3D (raw-parameter mode) needs scales and rotations, P filter values, and neither colors2 nor binning_capacity");
        if (!sh_second.is_none() && (colors2.has_value() || fixed || sh.numel() == 0 || colors.numel() != 0))
            throw std::runtime_error("sh_second (two tones of one SH block) needs SH colours and neither colors2 nor binning_capacity");
        if (colors2.has_value() && (!sh_tone.is_none() || fixed || sh.numel() != 0 || colors.numel() != 3 * P64 || colors2->numel() != 3 * P64))
            throw std::runtime_error("colors2 needs precomputed colours of P x 3 in both sets (no SH, no sh_tone, no binning_capacity)");
        if (fixed && debug) throw std::runtime_error("binning_capacity (the fixed-capacity forward) has no debug mode");
        // (the reference's binding checks none of its per-Gaussian arguments, rasterize_points.cu:59-61 is all there is; these cost nothing)
        auto sized = [&](const torch::Tensor& t, int64_t per, const char* name) {
            if (t.numel() != 0 && t.numel() != per * P64)
                throw std::runtime_error(std::string(name) + " must have " + std::to_string(per) + " * P elements");
        };
        sized(colors, 3, "colors_precomp"); sized(opacity, 1, "opacities"); sized(scales, 3, "scales"); sized(rotations, 4, "rotations");
        sized(cov3D_precomp, 6, "cov3D_precomp");
        if (sh.numel() != 0 && (sh.ndimension() != 3 || sh.size(0) != P64 || sh.size(2) != 3)) throw std::runtime_error("sh must have dimensions (P, M, 3)");
        if (opacity.numel() != P64) throw std::runtime_error("opacities must have P elements");
    }
    if (P == 0) {   // :83: nothing is launched, the image stays zero
        py::list out;
        out.append(0); out.append(torch::zeros({3, H, W}, f)); out.append(torch::zeros({0}, means3D.options().dtype(torch::kInt32)));
        out.append(geomBuffer); out.append(binningBuffer); out.append(imgBuffer);
        if (two) out.append(torch::zeros({3, H, W}, f));
        return py::tuple(out);
    }
    torch::Tensor out_color = torch::empty({3, H, W}, f), radii = torch::empty({P}, means3D.options().dtype(torch::kInt32)), out_color2;
    const auto m3 = f32(means3D, dev), bg = f32(background, dev), col = f32(colors, dev), op = f32(opacity, dev), sc = f32(scales, dev),
               rot = f32(rotations, dev), cov = f32(cov3D_precomp, dev), vm = f32(viewmatrix, dev), pm = f32(projmatrix, dev),
               cam = f32(campos, dev), shs = f32(sh, dev);
    const torch::Tensor so = subpixel_offset.has_value() ? f32(*subpixel_offset, dev) : torch::Tensor();
    wg_forward_args a{};
    a.struct_size = sizeof(a);
    a.geometry_alloc = resize_cb; a.geometry_user = &geomBuffer; a.binning_alloc = resize_cb; a.binning_user = &binningBuffer;
    a.image_alloc = resize_cb; a.image_user = &imgBuffer;
    a.P = P; a.D = degree; a.M = shs.numel() ? static_cast<int>(shs.size(1)) : 0; a.width = W; a.height = H; a.prefiltered = prefiltered; a.debug = debug;
    a.scale_modifier = scale_modifier; a.tan_fovx = tan_fovx; a.tan_fovy = tan_fovy; a.kernel_size = kernel_size;
    a.background = ptr(bg); a.means3D = ptr(m3); a.shs = ptr(shs); a.colors_precomp = ptr(col); a.opacities = ptr(op); a.scales = ptr(sc);
    a.rotations = ptr(rot); a.cov3D_precomp = ptr(cov); a.viewmatrix = ptr(vm); a.projmatrix = ptr(pm); a.cam_pos = ptr(cam);
    a.subpixel_offset = so.defined() ? ptr(so) : nullptr;
    a.out_color = out_color.data_ptr<float>(); a.radii = radii.data_ptr<int>();
    a.stream = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(dev.index()).stream();
    Tone tone, tone2;
    tone.parse(sh_tone, dev, P, false, f);
    if (tone.given) a.tone = &tone.t;
    wg_second_image second{};
    torch::Tensor c2;
    if (two) {
        out_color2 = torch::empty({3, H, W}, f);
        second.out_color2 = out_color2.data_ptr<float>();
        if (!sh_second.is_none()) {
            tone2.parse(sh_second, dev, P, false, f);
            a.tone2 = &tone2.t; a.sh_second = 1;
        } else {
            c2 = f32(*colors2, dev);
            second.colors_precomp2 = ptr(c2);
        }
        a.second = &second;
    }
    wg_raw_gaussians raw{};
    torch::Tensor f3d;
    if (filter_3D.has_value()) {
        f3d = f32(*filter_3D, dev);
        raw.filter_3D = ptr(f3d);
        a.raw = &raw;
    }
    if (!binning_capacity.is_none()) {
        a.binning_capacity = binning_capacity.cast<int>();
        if (a.binning_capacity <= 0) throw std::runtime_error("binning_capacity must be positive");
    }
    const wg_call_options co = call_options(options);
    a.options = &co;
    int rendered;
    {
        py::gil_scoped_release nogil;   // (the call may wait for the frame's instance count: other Python threads run meanwhile)
        rendered = wg_rasterize_forward_ex(&a);
    }
    check(rendered, "wg_rasterize_forward");
    py::list out;
    out.append(rendered); out.append(out_color); out.append(radii); out.append(geomBuffer); out.append(binningBuffer); out.append(imgBuffer);
    if (two) out.append(out_color2);
    return py::tuple(out);
}

py::tuple RasterizeGaussiansBackwardEx(const torch::Tensor& background, const torch::Tensor& means3D, const torch::Tensor& radii, const torch::Tensor& colors,
                                       const torch::Tensor& scales, const torch::Tensor& rotations, const float scale_modifier,
                                       const torch::Tensor& cov3D_precomp, const torch::Tensor& viewmatrix, const torch::Tensor& projmatrix,
                                       const float tan_fovx, const float tan_fovy, const float kernel_size, const OptTensor& subpixel_offset,
                                       const torch::Tensor& dL_dout_color, const torch::Tensor& sh, const int degree, const torch::Tensor& campos,
                                       const torch::Tensor& geomBuffer, const int R, const torch::Tensor& binningBuffer, const torch::Tensor& imageBuffer,
                                       const bool debug, const py::object& sh_tone, const OptTensor& dL_dout_color2, const py::object& raw_in,
                                       const py::object& sh_second, const std::tuple<int, int, int>& options) {
    const auto dev = means3D.device();
    const c10::DeviceGuard guard(dev);
    const int P = static_cast<int>(means3D.size(0));
    const int H = static_cast<int>(dL_dout_color.size(1)), W = static_cast<int>(dL_dout_color.size(2));
    const auto shs = f32(sh, dev);
    const int M = shs.numel() ? static_cast<int>(shs.size(1)) : 0;
    const auto f = means3D.options().dtype(torch::kFloat32);
    const wg_call_options co = call_options(options);
    const bool record = P != 0 && (co.grad_record != 0 || co.deterministic_backward != 0);
    const bool two_tone = !sh_second.is_none();
    const bool dual = dL_dout_color2.has_value() && !two_tone;
    if (dL_dout_color.ndimension() != 3 || dL_dout_color.size(0) != 3) throw std::runtime_error("dL_dout_color must have dimensions (3, H, W)");
    if (dL_dout_color2.has_value() && dL_dout_color2->numel() != dL_dout_color.numel()) throw std::runtime_error("dL_dout_color2 must have the first image's dimensions");
    if (P != 0 && (radii.numel() != (int64_t)P || (colors.numel() != 0 && colors.numel() != 3 * (int64_t)P) || (scales.numel() != 0 && scales.numel() != 3 * (int64_t)P) ||
                   (rotations.numel() != 0 && rotations.numel() != 4 * (int64_t)P) || (cov3D_precomp.numel() != 0 && cov3D_precomp.numel() != 6 * (int64_t)P) ||
                   (sh.numel() != 0 && (sh.ndimension() != 3 || sh.size(0) != (int64_t)P || sh.size(2) != 3))))
        throw std::runtime_error("a per-Gaussian argument of the backward call does not have P rows");
    // the reference zero-fills nine tensors (:157-165); with the gradient record every one of them is fully written by the library
    // (zeros for culled Gaussians), without it the four accumulation targets must arrive zeroed
    auto alloc = [&](std::initializer_list<int64_t> shape, bool accumulated) {
        return (P == 0 || (accumulated && !record)) ? torch::zeros(shape, f) : torch::empty(shape, f);
    };
    torch::Tensor dL_dmeans3D = alloc({P, 3}, false), dL_dmeans2D = alloc({P, 3}, true), dL_dcolors = alloc({P, 3}, true),
                  dL_dconic = record ? torch::Tensor() : torch::zeros({P, 2, 2}, f), dL_dopacity = alloc({P, 1}, true),
                  dL_dcov3D = alloc({P, 6}, false), dL_dsh = alloc({P, M, 3}, false);
    const bool have_scales = scales.numel() != 0;
    torch::Tensor dL_dscales = have_scales ? alloc({P, 3}, false) : torch::zeros({P, 3}, f),
                  dL_drotations = have_scales ? alloc({P, 4}, false) : torch::zeros({P, 4}, f);
    torch::Tensor dL_dcolors2 = dual ? alloc({P, 3}, false) : torch::Tensor();
    Tone tone, tone2;
    tone.parse(sh_tone, dev, P, true, f);
    if (two_tone) tone2.parse(sh_second, dev, P, true, f);
    if (P != 0) {
        const auto m3 = f32(means3D, dev), bg = f32(background, dev), col = f32(colors, dev), sc = f32(scales, dev), rot = f32(rotations, dev),
                   cov = f32(cov3D_precomp, dev), vm = f32(viewmatrix, dev), pm = f32(projmatrix, dev), cam = f32(campos, dev), dL = f32(dL_dout_color, dev);
        const torch::Tensor so = subpixel_offset.has_value() ? f32(*subpixel_offset, dev) : torch::Tensor();
        const auto rad = radii.contiguous();
        wg_backward_args a{};
        a.struct_size = sizeof(a);
        a.P = P; a.D = degree; a.M = M; a.R = R; a.width = W; a.height = H; a.debug = debug;
        a.scale_modifier = scale_modifier; a.tan_fovx = tan_fovx; a.tan_fovy = tan_fovy; a.kernel_size = kernel_size;
        a.background = ptr(bg); a.means3D = ptr(m3); a.shs = ptr(shs); a.colors_precomp = ptr(col); a.scales = ptr(sc); a.rotations = ptr(rot);
        a.cov3D_precomp = ptr(cov); a.viewmatrix = ptr(vm); a.projmatrix = ptr(pm); a.campos = ptr(cam);
        a.subpixel_offset = so.defined() ? ptr(so) : nullptr;
        a.radii = rad.numel() ? rad.data_ptr<int>() : nullptr;
        a.geom_buffer = reinterpret_cast<char*>(geomBuffer.data_ptr()); a.binning_buffer = reinterpret_cast<char*>(binningBuffer.data_ptr());
        a.image_buffer = reinterpret_cast<char*>(imageBuffer.data_ptr());
        a.dL_dpix = ptr(dL);
        a.dL_dmean2D = dL_dmeans2D.data_ptr<float>(); a.dL_dconic = dL_dconic.defined() ? dL_dconic.data_ptr<float>() : nullptr;
        a.dL_dopacity = dL_dopacity.data_ptr<float>(); a.dL_dcolor = dL_dcolors.data_ptr<float>(); a.dL_dmean3D = dL_dmeans3D.data_ptr<float>();
        a.dL_dcov3D = dL_dcov3D.data_ptr<float>(); a.dL_dsh = M ? dL_dsh.data_ptr<float>() : nullptr; a.dL_dscale = dL_dscales.data_ptr<float>();
        a.dL_drot = dL_drotations.data_ptr<float>();
        a.stream = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(dev.index()).stream();
        if (tone.given) a.tone = &tone.t;
        wg_second_image second{};
        torch::Tensor dL2;
        if (two_tone || dual) {
            if (!dL_dout_color2.has_value()) throw std::runtime_error("sh_second needs the second image's cotangent (dL_dout_color2)");
            dL2 = f32(*dL_dout_color2, dev);
            second.dL_dpix2 = ptr(dL2);
            if (two_tone) { a.tone2 = &tone2.t; a.sh_second = 1; } else second.dL_dcolor2 = dL_dcolors2.data_ptr<float>();
            a.second = &second;
        }
        wg_raw_gaussians raw{};
        torch::Tensor f3d, rop;
        if (!raw_in.is_none()) {
            const py::tuple tu = raw_in.cast<py::tuple>();
            if (tu.size() != 2) throw std::runtime_error("raw is (filter_3D, raw_opacities)");
            f3d = f32(tu[0].cast<torch::Tensor>(), dev);
            rop = f32(tu[1].cast<torch::Tensor>(), dev);
            if (f3d.numel() != (int64_t)P || rop.numel() != (int64_t)P) throw std::runtime_error("raw = (filter_3D, raw_opacities) must have P elements each");
            raw.filter_3D = ptr(f3d); raw.raw_opacities = ptr(rop);
            a.raw = &raw;
        }
        a.options = &co;
        int status;
        {
            py::gil_scoped_release nogil;   // (as the forward call: a deferred frame's verdict may be waited for here)
            status = wg_rasterize_backward_ex(&a);
        }
        check(status, "wg_rasterize_backward");
    }
    py::list out;
    for (const auto& t : {dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations}) out.append(t);   // :201
    if (two_tone) {
        out.append(tone.given ? tone.grad(tone.g_mul) : py::none()); out.append(tone.given ? tone.grad(tone.g_offset) : py::none());
        out.append(tone2.grad(tone2.g_mul)); out.append(tone2.grad(tone2.g_offset));
    } else if (dual) {
        out.append(dL_dcolors2);
    } else if (tone.given) {
        out.append(tone.grad(tone.g_mul)); out.append(tone.grad(tone.g_offset));
    }
    return py::tuple(out);
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {   // ext.cpp:15-19
    m.def("rasterize_gaussians", &RasterizeGaussiansHIP);
    m.def("rasterize_gaussians_backward", &RasterizeGaussiansBackwardHIP);
    m.def("mark_visible", &markVisible);
    m.def("rasterize_gaussians_ex", &RasterizeGaussiansEx);                    // + the blocks of wg_forward_args
    m.def("rasterize_gaussians_backward_ex", &RasterizeGaussiansBackwardEx);   // + the blocks of wg_backward_args
}
