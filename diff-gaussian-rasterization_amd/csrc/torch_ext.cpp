// torch_ext.cpp -- compiled `_C` of the light variant: the counterpart of the reference's pybind11 torch extension
// (L/ext.cpp:15-19, L/rasterize_points.cu:35-256) over the gfx950 C ABI (include/dgr_hip.h).
//
// torch supplies device memory, the current HIP stream and the device guard; every compute call goes through the C
// ABI in lib/libdgr_hip.so.  The Python side (dgr_amd/light.py) keeps only the policy that is cheap there -- the
// binning capacity learned per shape and the list of lazily checked status tickets -- and hands it in / gets it back
// as plain integers, so that a forward costs one pybind call instead of ~40 Python-level tensor operations and a
// 40-argument ctypes call (profiles/host_breakdown.py: 137 + 162 us per view in the ctypes binding).
#include <torch/extension.h>

// (a ROCm build of torch calls its devices "cuda": the guard and stream accessors that accept them are the
// "masquerading" ones)
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>

#include <stdexcept>
#include <string>
#include <tuple>
#include <vector>

#include "dgr_hip.h"

namespace {

using at::Tensor;

[[noreturn]] void fail(int rc) {
    const std::string msg = dgr_last_error();
    if (rc == DGR_ERR_PREFILTERED) throw std::runtime_error("Point is filtered although prefiltered is set. This shouldn't happen!");
    if (rc == DGR_ERR_BAD_ARGUMENT) throw std::runtime_error("dgr_hip: bad argument: " + msg);
    throw std::runtime_error("dgr_hip: error " + std::to_string(rc) + ": " + msg);
}
inline void check(long rc) {
    if (rc < 0) fail((int)rc);
}

// contiguous fp32 tensor on `dev` (L/rasterize_points.cu:101-125 calls .contiguous() on every input)
inline Tensor f32c(const Tensor& t, const c10::Device& dev) {
    if (t.scalar_type() == at::kFloat && t.is_contiguous() && t.device() == dev) return t;
    return t.to(dev, at::kFloat).contiguous();
}
// the reference's nullptr convention: an empty tensor stands for "None"
template <typename T>
inline T* ptr(const Tensor& t) {
    return t.numel() == 0 ? nullptr : t.data_ptr<T>();
}
inline char* bytes(const Tensor& t) { return t.numel() == 0 ? nullptr : reinterpret_cast<char*>(t.data_ptr()); }
inline void* stream_of(const c10::Device& dev) { return (void*)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(dev.index()).stream(); }

struct Alloc {
    Tensor* t;
    c10::Device dev;
};
char* resize_cb(size_t n, void* user) {  // the reference's resizeFunctional (L/rasterize_points.cu:27-33)
    auto* a = static_cast<Alloc*>(user);
    *a->t = at::empty({(long long)std::max<size_t>(n, 1)}, at::TensorOptions().dtype(at::kByte).device(a->dev));
    return reinterpret_cast<char*>(a->t->data_ptr());
}
// dgr_light_forward takes ONE user pointer for its three callbacks: three trampolines route to three tensors
struct Alloc3 {
    Alloc geom, binning, img;
};
char* cb_geom(size_t n, void* u) { return resize_cb(n, &static_cast<Alloc3*>(u)->geom); }
char* cb_binning(size_t n, void* u) { return resize_cb(n, &static_cast<Alloc3*>(u)->binning); }
char* cb_img(size_t n, void* u) { return resize_cb(n, &static_cast<Alloc3*>(u)->img); }

// mode: 0 = callback entry point (the strict mirror: allocation callbacks + the reference's blocking read),
//       1 = presized, strict: one host wait until num_rendered is known; retries a too-small capacity itself,
//       2 = presized, lazy: no host synchronisation; returns a status ticket (dgr_status_post) or -1 while capturing.
// Returns (num_rendered or -1, ticket or -1, capacity used, device status word, color, depth, median, var, alpha,
//          radii, geom, binning, img, gau_uncertainty, gau_related_pixels).
std::tuple<long, long, long, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor>
light_forward(const Tensor& background, const Tensor& means3D_, const Tensor& colors_, const Tensor& opacity_,
              const Tensor& scales_, const Tensor& rotations_, double scale_modifier, const Tensor& cov3D_,
              const Tensor& viewmatrix_, const Tensor& gt_depth_, const Tensor& projmatrix_, double tan_fovx,
              double tan_fovy, long H, long W, const Tensor& sh_, long degree, const Tensor& campos_, bool prefiltered,
              bool debug, long capacity, long mode) {
    if (means3D_.dim() != 2 || means3D_.size(1) != 3) throw std::runtime_error("means3D must have dimensions (num_points, 3)");
    const c10::Device dev = means3D_.device();
    if (!dev.is_cuda()) throw std::runtime_error("dgr_hip runs on the GPU only (no CPU path exists, as in the reference)");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(dev);
    const int P = (int)means3D_.size(0);
    const Tensor means3D = f32c(means3D_, dev), bg = f32c(background, dev), colors = f32c(colors_, dev),
                 opacity = f32c(opacity_, dev), scales = f32c(scales_, dev), rotations = f32c(rotations_, dev),
                 cov3D = f32c(cov3D_, dev), view = f32c(viewmatrix_, dev), proj = f32c(projmatrix_, dev),
                 campos = f32c(campos_, dev), gt = f32c(gt_depth_, dev), sh = f32c(sh_, dev);
    const int M = sh.numel() != 0 ? (int)sh.size(1) : 0;
    const auto f32 = at::TensorOptions().dtype(at::kFloat).device(dev);
    const auto i32 = at::TensorOptions().dtype(at::kInt).device(dev);
    const auto u8 = at::TensorOptions().dtype(at::kByte).device(dev);
    Tensor color = at::empty({3, H, W}, f32), depth = at::empty({1, H, W}, f32), median = at::empty({1, H, W}, f32),
           var = at::empty({1, H, W}, f32), alpha = at::empty({1, H, W}, f32);
    // radii is written for every Gaussian and the two median statistics are cleared by the kernels
    Tensor radii = P ? at::empty({P}, i32) : at::zeros({P}, i32);
    Tensor unc = P ? at::empty({P, 1}, f32) : at::zeros({P, 1}, f32);
    Tensor px = P ? at::empty({P, 1}, i32) : at::zeros({P, 1}, i32);
    void* st = stream_of(dev);
    Tensor geom, binning, img, status = at::empty({4}, i32);
    long rendered = -1, ticket = -1;

    if (mode == 0 || P == 0) {
        geom = at::empty({0}, u8); binning = at::empty({0}, u8); img = at::empty({0}, u8);
        Alloc3 al{{&geom, dev}, {&binning, dev}, {&img, dev}};
        const int rc = dgr_light_forward(st, cb_geom, cb_binning, cb_img, &al, P, (int)degree, M, ptr<float>(bg), (int)W, (int)H,
                                         ptr<float>(means3D), ptr<float>(sh), ptr<float>(colors), ptr<float>(opacity),
                                         ptr<float>(scales), (float)scale_modifier, ptr<float>(rotations), ptr<float>(cov3D),
                                         ptr<float>(view), ptr<float>(proj), ptr<float>(campos), (float)tan_fovx,
                                         (float)tan_fovy, prefiltered ? 1 : 0, ptr<float>(color), ptr<float>(depth),
                                         ptr<float>(median), ptr<float>(alpha), ptr<float>(gt), ptr<float>(var),
                                         ptr<float>(unc), ptr<int>(px), ptr<int>(radii), debug ? 1 : 0);
        check(rc);
        rendered = rc;
        return {rendered, ticket, rendered, status, color, depth, median, var, alpha, radii, geom, binning, img, unc, px};
    }
    geom = at::empty({(long long)dgr_geometry_bytes(P)}, u8);
    img = at::empty({(long long)dgr_image_bytes((int)W, (int)H)}, u8);
    auto run = [&](long cap) {
        binning = at::empty({(long long)dgr_binning_bytes((int)cap, (int)W, (int)H)}, u8);
        check(dgr_light_forward_presized(st, (char*)geom.data_ptr(), (char*)binning.data_ptr(), (int)cap, (char*)img.data_ptr(),
                                         status.data_ptr<int>(), P, (int)degree, M, ptr<float>(bg), (int)W, (int)H,
                                         ptr<float>(means3D), ptr<float>(sh), ptr<float>(colors), ptr<float>(opacity),
                                         ptr<float>(scales), (float)scale_modifier, ptr<float>(rotations), ptr<float>(cov3D),
                                         ptr<float>(view), ptr<float>(proj), ptr<float>(campos), (float)tan_fovx,
                                         (float)tan_fovy, prefiltered ? 1 : 0, ptr<float>(color), ptr<float>(depth),
                                         ptr<float>(median), ptr<float>(alpha), ptr<float>(gt), ptr<float>(var),
                                         ptr<float>(unc), ptr<int>(px), ptr<int>(radii)));
    };
    if (mode == 2) {
        run(capacity);
        if (!dgr_stream_is_capturing(st)) {
            ticket = dgr_status_post(st, status.data_ptr<int>());
            check(ticket);
        }
        return {rendered, ticket, capacity, status, color, depth, median, var, alpha, radii, geom, binning, img, unc, px};
    }
    long cap = capacity;
    for (;;) {
        check(dgr_early_status_arm());
        run(cap);
        int s[4] = {0, 0, 0, 0};
        check(dgr_early_status_wait(s));  // the one host wait of this forward: until num_rendered is known
        if (s[2]) throw std::runtime_error("Point is filtered although prefiltered is set. This shouldn't happen!");
        rendered = s[0];
        if (rendered <= cap) break;
        cap = (long)(rendered * 1.1) + 4096;  // overflow: every tile list was left empty; run again
    }
    if (debug) check(hipStreamSynchronize((hipStream_t)st) == hipSuccess ? 0 : DGR_ERR_HIP);
    return {rendered, ticket, cap, status, color, depth, median, var, alpha, radii, geom, binning, img, unc, px};
}

// L/rasterize_points.cu:131-236.  Returns (dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales,
// dL_drotations, dL_dview [1,4,4]); the first eight are views of one flat arena whose leading segments (means3D, means2D,
// sh, opacity, scales, rotations) form the multi-GPU all-reduce payload, or undefined tensors (None) when
// need_gaussian_grads is false (tracking: the library then skips every dense per-Gaussian row).
std::vector<Tensor> light_backward(const Tensor& background, const Tensor& means3D_, const Tensor& radii, const Tensor& colors_,
                                   const Tensor& scales_, const Tensor& rotations_, double scale_modifier, const Tensor& cov3D_,
                                   const Tensor& viewmatrix_, const Tensor& projmatrix_, double tan_fovx, double tan_fovy,
                                   const Tensor& dL_dout_color, const Tensor& dL_dout_depth, const Tensor& dL_dout_median,
                                   const Tensor& dL_dout_var, const Tensor& gt_depth_, const Tensor& sh_, long degree,
                                   const Tensor& campos_, const Tensor& geomBuffer, long R, const Tensor& binningBuffer,
                                   const Tensor& imageBuffer, const Tensor& alphas_, bool debug, const Tensor& perspec_,
                                   bool track_off, bool map_off, bool need_gaussian_grads) {
    const c10::Device dev = means3D_.device();
    c10::hip::HIPGuardMasqueradingAsCUDA guard(dev);
    const int P = (int)means3D_.size(0);
    const long H = dL_dout_color.size(1), W = dL_dout_color.size(2);
    const Tensor means3D = f32c(means3D_, dev), bg = f32c(background, dev), colors = f32c(colors_, dev),
                 scales = f32c(scales_, dev), rotations = f32c(rotations_, dev), cov3D = f32c(cov3D_, dev),
                 view = f32c(viewmatrix_, dev), proj = f32c(projmatrix_, dev), campos = f32c(campos_, dev),
                 gt = f32c(gt_depth_, dev), sh = f32c(sh_, dev), alphas = f32c(alphas_, dev), perspec = f32c(perspec_, dev),
                 gC = f32c(dL_dout_color, dev), gD = f32c(dL_dout_depth, dev), gM = f32c(dL_dout_median, dev),
                 gV = f32c(dL_dout_var, dev);
    const int M = sh.numel() != 0 ? (int)sh.size(1) : 0;
    const auto f32 = at::TensorOptions().dtype(at::kFloat).device(dev);
    std::vector<Tensor> g(9);
    float* gp[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    if (need_gaussian_grads) {
        // arena segment order (dgr_amd.light._grad_arena): means3D, means2D, sh, opacity, scales, rotations | cov3D, colors
        const long long n[8] = {3LL * P, 3LL * P, 3LL * M * P, P, 3LL * P, 4LL * P, 6LL * P, 3LL * P};
        long long off[8], o = 0;
        for (int i = 0; i < 8; i++) { off[i] = o; o += (n[i] + 63) / 64 * 64; }  // 256-byte aligned segments
        Tensor arena = P ? at::empty({std::max<long long>(o, 1)}, f32) : at::zeros({std::max<long long>(o, 1)}, f32);
        auto seg = [&](int i, c10::IntArrayRef shape) { return arena.narrow(0, off[i], n[i]).view(shape); };
        const Tensor dmeans3D = seg(0, {P, 3}), dmeans2D = seg(1, {P, 3}), dsh = seg(2, {P, M, 3}), dop = seg(3, {P, 1}),
                     dsc = seg(4, {P, 3}), drot = seg(5, {P, 4}), dcov = seg(6, {P, 6}), dcol = seg(7, {P, 3});
        // return order of the reference binding: means2D, colors, opacity, means3D, cov3D, sh, scales, rotations
        g[0] = dmeans2D; g[1] = dcol; g[2] = dop; g[3] = dmeans3D; g[4] = dcov; g[5] = dsh; g[6] = dsc; g[7] = drot;
        for (int i = 0; i < 8; i++) gp[i] = ptr<float>(g[i]);
    } else {
        map_off = true;  // nobody reads the per-Gaussian sums: the blend kernel forms the three pose sums only
    }
    Tensor dview = at::empty({1, 4, 4}, f32);  // [1,4,4]: what L/__init__.py:160-161 sums over dim 0
    Tensor scratch = at::empty({(long long)std::max<size_t>(dgr_light_backward_scratch_bytes(P, (int)W, (int)H), 1)},
                               at::TensorOptions().dtype(at::kByte).device(dev));
    check(dgr_light_backward(stream_of(dev), P, (int)degree, M, (int)R, ptr<float>(bg), (int)W, (int)H, ptr<float>(means3D),
                             ptr<float>(sh), ptr<float>(colors), ptr<float>(alphas), ptr<float>(scales), (float)scale_modifier,
                             ptr<float>(rotations), ptr<float>(cov3D), ptr<float>(view), ptr<float>(proj), ptr<float>(campos),
                             (float)tan_fovx, (float)tan_fovy, ptr<int>(radii), bytes(geomBuffer),
                             bytes(binningBuffer), bytes(imageBuffer), ptr<float>(gC),
                             ptr<float>(gD), ptr<float>(gM), ptr<float>(gV), gp[0], nullptr, gp[2], gp[1], nullptr, gp[3], gp[4],
                             gp[5], gp[6], gp[7], debug ? 1 : 0, nullptr, ptr<float>(perspec), dview.data_ptr<float>(), nullptr,
                             ptr<float>(gt), track_off ? 1 : 0, map_off ? 1 : 0, (char*)scratch.data_ptr(), (size_t)scratch.numel()));
    g[8] = dview;
    return g;
}


// ------------------------------------------------------------------------------------------------ full variant
// F/rasterize_points.cu:35-120.  Modes as light_forward.  Returns (num_rendered or -1, num_related or -1, ticket or -1,
// capacity used, device status word, color, depth, uncertainty, radii, geom, binning, img).
std::tuple<long, long, long, long, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor>
full_forward(const Tensor& background, const Tensor& means3D_, const Tensor& colors_, const Tensor& opacity_, const Tensor& scales_,
             const Tensor& rotations_, double scale_modifier, const Tensor& cov3D_, const Tensor& viewmatrix_,
             const Tensor& gt_depth_, const Tensor& projmatrix_, double tan_fovx, double tan_fovy, long H, long W,
             const Tensor& sh_, long degree, const Tensor& campos_, bool prefiltered, long capacity, long mode) {
    if (means3D_.dim() != 2 || means3D_.size(1) != 3) throw std::runtime_error("means3D must have dimensions (num_points, 3)");
    const c10::Device dev = means3D_.device();
    if (!dev.is_cuda()) throw std::runtime_error("dgr_hip runs on the GPU only (no CPU path exists, as in the reference)");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(dev);
    const int P = (int)means3D_.size(0);
    const Tensor means3D = f32c(means3D_, dev), bg = f32c(background, dev), colors = f32c(colors_, dev),
                 opacity = f32c(opacity_, dev), scales = f32c(scales_, dev), rotations = f32c(rotations_, dev),
                 cov3D = f32c(cov3D_, dev), view = f32c(viewmatrix_, dev), proj = f32c(projmatrix_, dev),
                 campos = f32c(campos_, dev), gt = f32c(gt_depth_, dev), sh = f32c(sh_, dev);
    const int M = sh.numel() != 0 ? (int)sh.size(1) : 0;
    const auto f32 = at::TensorOptions().dtype(at::kFloat).device(dev);
    const auto i32 = at::TensorOptions().dtype(at::kInt).device(dev);
    const auto u8 = at::TensorOptions().dtype(at::kByte).device(dev);
    Tensor color = at::empty({3, H, W}, f32), depth = at::empty({1, H, W}, f32), unc = at::empty({1, H, W}, f32);
    Tensor radii = at::zeros({P}, i32);
    void* st = stream_of(dev);
    Tensor geom, binning, img, status = at::empty({4}, i32);
    long rendered = -1, related = -1, ticket = -1;
    if (mode == 0 || P == 0) {
        geom = at::empty({0}, u8); binning = at::empty({0}, u8); img = at::empty({0}, u8);
        Alloc3 al{{&geom, dev}, {&binning, dev}, {&img, dev}};
        int ng = 0;
        const int rc = dgr_full_forward(st, cb_geom, cb_binning, cb_img, &al, P, (int)degree, M, ptr<float>(bg), (int)W, (int)H,
                                        ptr<float>(means3D), ptr<float>(sh), ptr<float>(colors), ptr<float>(opacity),
                                        ptr<float>(scales), (float)scale_modifier, ptr<float>(rotations), ptr<float>(cov3D),
                                        ptr<float>(view), ptr<float>(proj), ptr<float>(campos), (float)tan_fovx, (float)tan_fovy,
                                        prefiltered ? 1 : 0, ptr<float>(color), ptr<float>(depth), ptr<float>(gt), ptr<float>(unc),
                                        ptr<int>(radii), &ng);
        check(rc);
        return {rc, ng, ticket, rc, status, color, depth, unc, radii, geom, binning, img};
    }
    geom = at::empty({(long long)dgr_geometry_bytes(P)}, u8);
    img = at::empty({(long long)dgr_image_bytes((int)W, (int)H)}, u8);
    auto run = [&](long cap) {
        binning = at::empty({(long long)dgr_binning_bytes((int)cap, (int)W, (int)H)}, u8);
        check(dgr_full_forward_presized(st, (char*)geom.data_ptr(), (char*)binning.data_ptr(), (int)cap, (char*)img.data_ptr(),
                                        status.data_ptr<int>(), P, (int)degree, M, ptr<float>(bg), (int)W, (int)H,
                                        ptr<float>(means3D), ptr<float>(sh), ptr<float>(colors), ptr<float>(opacity),
                                        ptr<float>(scales), (float)scale_modifier, ptr<float>(rotations), ptr<float>(cov3D),
                                        ptr<float>(view), ptr<float>(proj), ptr<float>(campos), (float)tan_fovx, (float)tan_fovy,
                                        prefiltered ? 1 : 0, ptr<float>(color), ptr<float>(depth), ptr<float>(gt), ptr<float>(unc),
                                        ptr<int>(radii)));
    };
    if (mode == 2) {
        run(capacity);
        if (!dgr_stream_is_capturing(st)) {
            ticket = dgr_status_post(st, status.data_ptr<int>());
            check(ticket);
        }
        return {rendered, related, ticket, capacity, status, color, depth, unc, radii, geom, binning, img};
    }
    long cap = capacity;
    for (;;) {
        check(dgr_early_status_arm());
        run(cap);
        int s[4] = {0, 0, 0, 0};
        check(dgr_early_status_wait(s));
        if (s[2]) throw std::runtime_error("Point is filtered although prefiltered is set. This shouldn't happen!");
        rendered = s[0];
        if (rendered <= cap) break;
        cap = (long)(rendered * 1.1) + 4096;
    }
    // num_related (the reference's NG) is produced by the forward blend: the reference's second blocking read
    // (F/cuda_rasterizer/rasterizer_impl.cu:498)
    related = status.to(at::kCPU).data_ptr<int>()[3];
    return {rendered, related, ticket, cap, status, color, depth, unc, radii, geom, binning, img};
}

// F/rasterize_points.cu:122-239; returns the nine gradients in the reference's order, dL_dview as [4,4]
std::vector<Tensor> full_backward(const Tensor& background, const Tensor& means3D_, const Tensor& radii, const Tensor& colors_,
                                  const Tensor& scales_, const Tensor& rotations_, double scale_modifier, const Tensor& cov3D_,
                                  const Tensor& viewmatrix_, const Tensor& gt_depth_, const Tensor& projmatrix_, double tan_fovx,
                                  double tan_fovy, const Tensor& dL_dout_color, const Tensor& dL_dout_depth,
                                  const Tensor& dL_dout_unc, const Tensor& sh_, long degree, const Tensor& campos_,
                                  const Tensor& geomBuffer, long R, const Tensor& binningBuffer, const Tensor& imageBuffer,
                                  long NG, const Tensor& perspec_, bool need_gaussian_grads) {
    (void)NG;
    const c10::Device dev = means3D_.device();
    c10::hip::HIPGuardMasqueradingAsCUDA guard(dev);
    const int P = (int)means3D_.size(0);
    const long H = dL_dout_color.size(1), W = dL_dout_color.size(2);
    const Tensor means3D = f32c(means3D_, dev), bg = f32c(background, dev), colors = f32c(colors_, dev),
                 scales = f32c(scales_, dev), rotations = f32c(rotations_, dev), cov3D = f32c(cov3D_, dev),
                 view = f32c(viewmatrix_, dev), proj = f32c(projmatrix_, dev), campos = f32c(campos_, dev),
                 gt = f32c(gt_depth_, dev), sh = f32c(sh_, dev), perspec = f32c(perspec_, dev), gC = f32c(dL_dout_color, dev),
                 gD = f32c(dL_dout_depth, dev), gU = f32c(dL_dout_unc, dev);
    const int M = sh.numel() != 0 ? (int)sh.size(1) : 0;
    const auto f32 = at::TensorOptions().dtype(at::kFloat).device(dev);
    std::vector<Tensor> g(9);
    float* gp[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    if (need_gaussian_grads) {
        const long long n[8] = {3LL * P, 3LL * P, 3LL * M * P, P, 3LL * P, 4LL * P, 6LL * P, 3LL * P};
        long long off[8], o = 0;
        for (int i = 0; i < 8; i++) { off[i] = o; o += (n[i] + 63) / 64 * 64; }
        Tensor arena = P ? at::empty({std::max<long long>(o, 1)}, f32) : at::zeros({std::max<long long>(o, 1)}, f32);
        auto seg = [&](int i, c10::IntArrayRef shape) { return arena.narrow(0, off[i], n[i]).view(shape); };
        g[3] = seg(0, {P, 3}); g[0] = seg(1, {P, 3}); g[5] = seg(2, {P, M, 3}); g[2] = seg(3, {P, 1});
        g[6] = seg(4, {P, 3}); g[7] = seg(5, {P, 4}); g[4] = seg(6, {P, 6}); g[1] = seg(7, {P, 3});
        for (int i = 0; i < 8; i++) gp[i] = ptr<float>(g[i]);
    }
    Tensor dview = at::empty({4, 4}, f32);
    Tensor scratch = at::empty({(long long)std::max<size_t>(dgr_light_backward_scratch_bytes(P, (int)W, (int)H), 1)},
                               at::TensorOptions().dtype(at::kByte).device(dev));
    // gp: [0] means2D [1] colors [2] opacity [3] means3D [4] cov3D [5] sh [6] scales [7] rotations
    check(dgr_full_backward(stream_of(dev), P, (int)degree, M, (int)R, ptr<float>(bg), (int)W, (int)H, ptr<float>(means3D),
                            ptr<float>(sh), ptr<float>(colors), ptr<float>(scales), (float)scale_modifier, ptr<float>(rotations),
                            ptr<float>(cov3D), ptr<float>(view), ptr<float>(proj), ptr<float>(campos), (float)tan_fovx,
                            (float)tan_fovy, ptr<int>(radii), bytes(geomBuffer), bytes(binningBuffer), bytes(imageBuffer),
                            ptr<float>(gC), ptr<float>(gD), gp[0], nullptr, gp[2], gp[1], gp[3], gp[4], gp[5], gp[6], gp[7], nullptr,
                            nullptr, nullptr, nullptr, nullptr, ptr<float>(perspec), nullptr, nullptr, nullptr,
                            dview.data_ptr<float>(), nullptr, nullptr, nullptr, ptr<float>(gt), ptr<float>(gU),
                            (char*)scratch.data_ptr(), (size_t)scratch.numel()));
    g[8] = dview;
    return g;
}

// ------------------------------------------------------------------------------------------------ batched views
// dgr_amd.batch over the C ABI's batched entry points (include/dgr_hip.h: dgr_light_forward_batch / _backward_batch): V
// cameras over one set of Gaussians per call.  ONE attempt with the given binning capacity per view and no host
// synchronisation; the capacity policy, the strict mode's status read and its retry stay in Python (dgr_amd/batch.py).
// post_status: copy every view's status word to pinned memory behind an event (lazy mode) and return the tickets.
// Returns ([V,4] status, color, depth, median, var, alpha, radii, geom, binning, img, unc, px) and the tickets.
inline char* row_bytes(const Tensor& t, long v) { return t.numel() == 0 ? nullptr : reinterpret_cast<char*>(t.data_ptr()) + v * t.stride(0) * t.element_size(); }
template <typename T>
inline T* row(const Tensor& t, long v) { return reinterpret_cast<T*>(row_bytes(t, v)); }

std::tuple<std::vector<Tensor>, std::vector<long>>
light_forward_batch(const Tensor& background, const Tensor& means3D_, const Tensor& colors_, const Tensor& opacity_,
                    const Tensor& scales_, const Tensor& rotations_, double scale_modifier, const Tensor& cov3D_,
                    const Tensor& viewmatrices_, const Tensor& gt_depths_, const Tensor& projmatrices_, double tan_fovx,
                    double tan_fovy, long H, long W, const Tensor& sh_, long degree, const Tensor& campos_, bool prefiltered,
                    long capacity, bool post_status) {
    if (means3D_.dim() != 2 || means3D_.size(1) != 3) throw std::runtime_error("means3D must have dimensions (num_points, 3)");
    const c10::Device dev = means3D_.device();
    if (!dev.is_cuda()) throw std::runtime_error("dgr_hip runs on the GPU only (no CPU path exists, as in the reference)");
    const long V = viewmatrices_.dim() == 3 ? viewmatrices_.size(0) : 0;
    if (V < 1 || V > DGR_MAX_BATCH_VIEWS) throw std::runtime_error("1 .. " + std::to_string(DGR_MAX_BATCH_VIEWS) + " views per batch");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(dev);
    const int P = (int)means3D_.size(0);
    const Tensor means3D = f32c(means3D_, dev), bg = f32c(background, dev), colors = f32c(colors_, dev),
                 opacity = f32c(opacity_, dev), scales = f32c(scales_, dev), rotations = f32c(rotations_, dev),
                 cov3D = f32c(cov3D_, dev), views = f32c(viewmatrices_, dev), projs = f32c(projmatrices_, dev),
                 campos = f32c(campos_, dev), gts = f32c(gt_depths_, dev), sh = f32c(sh_, dev);
    const int M = sh.numel() != 0 ? (int)sh.size(1) : 0;
    const auto f32 = at::TensorOptions().dtype(at::kFloat).device(dev);
    const auto i32 = at::TensorOptions().dtype(at::kInt).device(dev);
    const auto u8 = at::TensorOptions().dtype(at::kByte).device(dev);
    Tensor color = at::empty({V, 3, H, W}, f32), depth = at::empty({V, 1, H, W}, f32), median = at::empty({V, 1, H, W}, f32),
           var = at::empty({V, 1, H, W}, f32), alpha = at::empty({V, 1, H, W}, f32);
    Tensor radii = P ? at::empty({V, P}, i32) : at::zeros({V, P}, i32);
    Tensor unc = P ? at::empty({V, P, 1}, f32) : at::zeros({V, P, 1}, f32);
    Tensor px = P ? at::empty({V, P, 1}, i32) : at::zeros({V, P, 1}, i32);
    Tensor geom = at::empty({V, (long long)std::max<size_t>(dgr_geometry_bytes(P), 1)}, u8);
    Tensor img = at::empty({V, (long long)std::max<size_t>(dgr_image_bytes((int)W, (int)H), 1)}, u8);
    Tensor binning = at::empty({V, (long long)std::max<size_t>(dgr_binning_bytes((int)capacity, (int)W, (int)H), 1)}, u8);
    Tensor status = at::zeros({V, 4}, i32);
    dgr_light_view w[DGR_MAX_BATCH_VIEWS];
    for (long v = 0; v < V; v++) {
        w[v] = dgr_light_view{row_bytes(geom, v), row_bytes(binning, v), (int)capacity, row_bytes(img, v), row<int>(status, v),
                              row<float>(views, v), row<float>(projs, v), row<float>(campos, v), row<float>(color, v),
                              row<float>(depth, v), row<float>(median, v), row<float>(alpha, v), row<float>(gts, v),
                              row<float>(var, v), row<float>(unc, v), row<int>(px, v), row<int>(radii, v)};
    }
    void* st = stream_of(dev);
    check(dgr_light_forward_batch(st, (int)V, w, P, (int)degree, M, ptr<float>(bg), (int)W, (int)H, ptr<float>(means3D),
                                  ptr<float>(sh), ptr<float>(colors), ptr<float>(opacity), ptr<float>(scales), (float)scale_modifier,
                                  ptr<float>(rotations), ptr<float>(cov3D), (float)tan_fovx, (float)tan_fovy, prefiltered ? 1 : 0));
    std::vector<long> tickets;
    if (post_status && P > 0 && !dgr_stream_is_capturing(st)) {
        for (long v = 0; v < V; v++) {
            const long t = dgr_status_post(st, row<int>(status, v));
            check(t);
            tickets.push_back(t);
        }
    }
    return {{status, color, depth, median, var, alpha, radii, geom, binning, img, unc, px}, tickets};
}

// Returns (dL_dmeans2D [V,P,3] or None, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations --
// the SUMS over the views, views of one flat arena laid out as light_backward's -- and dL_dview [V,4,4]).
std::vector<Tensor> light_backward_batch(const Tensor& background, const Tensor& means3D_, const Tensor& radii, const Tensor& colors_,
                                         const Tensor& scales_, const Tensor& rotations_, double scale_modifier, const Tensor& cov3D_,
                                         const Tensor& viewmatrices_, const Tensor& projmatrices_, double tan_fovx, double tan_fovy,
                                         const Tensor& dL_dout_color, const Tensor& dL_dout_depth, const Tensor& dL_dout_median,
                                         const Tensor& dL_dout_var, const Tensor& gt_depths_, const Tensor& sh_, long degree,
                                         const Tensor& campos_, const Tensor& geom, const Tensor& binning, const Tensor& img,
                                         const Tensor& alphas_, const Tensor& perspec_, bool track_off, bool map_off,
                                         bool need_gaussian_grads, bool need_means2D) {
    const c10::Device dev = means3D_.device();
    c10::hip::HIPGuardMasqueradingAsCUDA guard(dev);
    const int P = (int)means3D_.size(0);
    const long V = viewmatrices_.size(0), H = dL_dout_color.size(2), W = dL_dout_color.size(3);
    if (V < 1 || V > DGR_MAX_BATCH_VIEWS) throw std::runtime_error("1 .. " + std::to_string(DGR_MAX_BATCH_VIEWS) + " views per batch");
    const Tensor means3D = f32c(means3D_, dev), bg = f32c(background, dev), colors = f32c(colors_, dev),
                 scales = f32c(scales_, dev), rotations = f32c(rotations_, dev), cov3D = f32c(cov3D_, dev),
                 views = f32c(viewmatrices_, dev), projs = f32c(projmatrices_, dev), campos = f32c(campos_, dev),
                 gts = f32c(gt_depths_, dev), sh = f32c(sh_, dev), alphas = f32c(alphas_, dev), perspec = f32c(perspec_, dev),
                 gC = f32c(dL_dout_color, dev), gD = f32c(dL_dout_depth, dev), gM = f32c(dL_dout_median, dev),
                 gV = f32c(dL_dout_var, dev);
    const int M = sh.numel() != 0 ? (int)sh.size(1) : 0;
    const auto f32 = at::TensorOptions().dtype(at::kFloat).device(dev);
    std::vector<Tensor> g(9);
    float* gp[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    Tensor d2;
    if (need_gaussian_grads) {
        const long long n[8] = {3LL * P, 3LL * P, 3LL * M * P, P, 3LL * P, 4LL * P, 6LL * P, 3LL * P};
        long long off[8], o = 0;
        for (int i = 0; i < 8; i++) { off[i] = o; o += (n[i] + 63) / 64 * 64; }
        Tensor arena = P ? at::empty({std::max<long long>(o, 1)}, f32) : at::zeros({std::max<long long>(o, 1)}, f32);
        auto seg = [&](int i, c10::IntArrayRef shape) { return arena.narrow(0, off[i], n[i]).view(shape); };
        seg(1, {P, 3}).zero_();  // the arena's one-view means2D slot: a batch returns those gradients per view, beside the arena
        g[3] = seg(0, {P, 3}); g[5] = seg(2, {P, M, 3}); g[2] = seg(3, {P, 1});
        g[6] = seg(4, {P, 3}); g[7] = seg(5, {P, 4}); g[4] = seg(6, {P, 6}); g[1] = seg(7, {P, 3});
        for (int i = 1; i < 8; i++) gp[i] = ptr<float>(g[i]);
        if (need_means2D) { d2 = at::empty({V, P, 3}, f32); g[0] = d2; }
    } else {
        map_off = true;  // nobody reads the per-Gaussian sums: the blend kernels form the three pose sums only
    }
    Tensor dview = at::empty({V, 4, 4}, f32);
    const size_t nscr = std::max<size_t>(dgr_light_backward_scratch_bytes(P, (int)W, (int)H), 1);
    Tensor scratch = at::empty({V, (long long)nscr}, at::TensorOptions().dtype(at::kByte).device(dev));
    dgr_light_view_grad w[DGR_MAX_BATCH_VIEWS];
    for (long v = 0; v < V; v++) {
        w[v] = dgr_light_view_grad{row_bytes(geom, v), row_bytes(binning, v), row_bytes(img, v), row<float>(views, v),
                                   row<float>(projs, v), row<float>(campos, v), ptr<float>(perspec), row<float>(alphas, v),
                                   row<float>(gts, v), row<int>(radii, v), row<float>(gC, v), row<float>(gD, v), row<float>(gM, v),
                                   row<float>(gV, v), d2.defined() ? row<float>(d2, v) : nullptr, row<float>(dview, v),
                                   row_bytes(scratch, v), nscr};
    }
    // gp: [1] colors [2] opacity [3] means3D [4] cov3D [5] sh [6] scales [7] rotations
    check(dgr_light_backward_batch(stream_of(dev), (int)V, w, P, (int)degree, M, ptr<float>(bg), (int)W, (int)H, ptr<float>(means3D),
                                   ptr<float>(sh), ptr<float>(colors), ptr<float>(scales), (float)scale_modifier,
                                   ptr<float>(rotations), ptr<float>(cov3D), (float)tan_fovx, (float)tan_fovy, gp[2], gp[1], gp[3],
                                   gp[4], gp[5], gp[6], gp[7], track_off ? 1 : 0, map_off ? 1 : 0));
    g[8] = dview;
    return g;
}

Tensor mark_visible(const Tensor& means3D_, const Tensor& viewmatrix_, const Tensor& projmatrix_) {  // L/rasterize_points.cu:238-256
    const c10::Device dev = means3D_.device();
    c10::hip::HIPGuardMasqueradingAsCUDA guard(dev);
    const int P = (int)means3D_.size(0);
    Tensor present = at::zeros({P}, at::TensorOptions().dtype(at::kBool).device(dev));
    if (P != 0) {
        const Tensor m = f32c(means3D_, dev), v = f32c(viewmatrix_, dev), pj = f32c(projmatrix_, dev);
        check(dgr_mark_visible(stream_of(dev), P, ptr<float>(m), ptr<float>(v), ptr<float>(pj),
                               reinterpret_cast<uint8_t*>(present.data_ptr<bool>())));
    }
    return present;
}

// status ticket of a lazy forward: None while the copy has not landed (wait = false), else [num_rendered, overflow,
// prefiltered violation, num_related]
py::object status_poll(long ticket, bool wait) {
    int s[4];
    const int rc = dgr_status_poll(ticket, wait ? 1 : 0, s);
    check(rc);
    if (rc == 0) return py::none();
    return py::make_tuple(s[0], s[1], s[2], s[3]);
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.def("light_forward", &light_forward);
    m.def("light_backward", &light_backward);
    m.def("full_forward", &full_forward);
    m.def("full_backward", &full_backward);
    m.def("light_forward_batch", &light_forward_batch);
    m.def("light_backward_batch", &light_backward_batch);
    m.def("mark_visible", &mark_visible);
    m.def("status_poll", &status_poll);
}
