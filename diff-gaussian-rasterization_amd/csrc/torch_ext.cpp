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
#include <ATen/hip/impl/HIPCachingAllocatorMasqueradingAsCUDA.h>

#include <atomic>
#include <chrono>
#include <mutex>
#include <stdexcept>
#include <string>
#include <tuple>
#include <vector>

#include "dgr_hip.h"

namespace {

using at::Tensor;

// ---- host-side profile of the binding (DGR_HOST_PROF=1; profiles/host_breakdown.py prints it): where a forward's and a
// backward's microseconds on the issuing thread go.  Off: one predictable branch per probe.
const bool g_host_prof = [] { const char* e = getenv("DGR_HOST_PROF"); return e && e[0] == '1'; }();
struct HostProf {
    const char* name;
    double us = 0;
    long n = 0;
};
HostProf g_hp[] = {{"fwd: apply() total"}, {"fwd: node forward()"}, {"fwd: core: guard + f32c"}, {"fwd: core: output allocations"},
                   {"fwd: core: status arm"}, {"fwd: core: state allocation"}, {"fwd: core: C ABI (launches)"},
                   {"fwd: save_for_backward + saved_data"}, {"bwd: node backward()"}, {"bwd: arena + scratch + dview"},
                   {"bwd: C ABI (launches)"}, {"bwd: unpack saved"}};
enum { HP_APPLY, HP_FWD, HP_PRELUDE, HP_OUT_ALLOC, HP_ARM, HP_STATE_ALLOC, HP_FWD_C, HP_SAVE, HP_BWD, HP_BWD_ALLOC, HP_BWD_C, HP_UNPACK };
struct Probe {
    int id;
    std::chrono::steady_clock::time_point t0;
    explicit Probe(int i) : id(g_host_prof ? i : -1) { if (id >= 0) t0 = std::chrono::steady_clock::now(); }
    void stop() {
        if (id < 0) return;
        g_hp[id].us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        g_hp[id].n++;
        id = -1;
    }
    ~Probe() { stop(); }
};
std::string host_prof_dump(bool reset) {
    std::string out;
    for (auto& h : g_hp) {
        if (h.n) out += std::string(h.name) + ": " + std::to_string(h.us / (double)h.n) + " us x " + std::to_string(h.n) + "\n";
        if (reset) { h.us = 0; h.n = 0; }
    }
    return out;
}

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
    // (an empty tensor stands for "None" and is passed on as nullptr: converting the caller's empty CPU tensor to the
    //  device would cost two dispatcher calls and an allocation per argument per call)
    if (t.numel() == 0 || (t.scalar_type() == at::kFloat && t.is_contiguous() && t.device() == dev)) return t;
    return t.to(dev, at::kFloat).contiguous();
}
// perspec_matrix: the kernels read entries 0 and 5 only (L/cr/backward.cu:725-739) -- the diagonal, which a transposed
// 4x4 (how callers usually hold Proj^T: `projection.transpose(0, 1)`) keeps in place: no copy kernel per backward for it
inline Tensor f32c_diag4(const Tensor& t, const c10::Device& dev) {
    if (t.scalar_type() == at::kFloat && t.device() == dev && t.dim() == 2 && t.size(0) == 4 && t.size(1) == 4 && t.stride(0) == 1 &&
        t.stride(1) == 4)
        return t;
    return f32c(t, dev);
}
// the reference's nullptr convention: an empty tensor stands for "None"
template <typename T>
inline T* ptr(const Tensor& t) {
    return t.numel() == 0 ? nullptr : t.data_ptr<T>();
}
inline char* bytes(const Tensor& t) { return t.numel() == 0 ? nullptr : reinterpret_cast<char*>(t.data_ptr()); }
inline void* stream_of(const c10::Device& dev) { return (void*)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(dev.index()).stream(); }

// PyTorch's rule for a tensor read on another stream than the one it was allocated on: tell the caching allocator, or
// the block is handed out again while that stream's kernels still read it.  A forward issued on a side stream (views in
// flight, dgr_amd.multiview.ViewStreams) does it for its inputs -- a per-view viewmatrix or gt_depth made on the
// caller's stream and dropped right after the call is the case that bites; the saved inputs are read by the backward
// on the same stream, so the one record covers both.  The ORIGINAL arguments are recorded: where f32c converts one (a
// transposed perspec_matrix is the usual case) the conversion reads it on this stream and its result is a temporary of
// this stream.  On the default stream: one comparison.
const bool g_record_inputs = [] { const char* e = getenv("DGR_RECORD_INPUT_STREAMS"); return !(e && e[0] == '0'); }();
inline void keep_until_read(const c10::Device& dev, std::initializer_list<const Tensor*> inputs) {
    if (!g_record_inputs) return;  // (DGR_RECORD_INPUT_STREAMS=0: the caller keeps its inputs alive until the streams are joined)
    const auto s = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(dev.index());
    if (s == c10::hip::getDefaultHIPStreamMasqueradingAsCUDA(dev.index())) return;
    for (const Tensor* t : inputs)
        if (t->defined() && t->numel() != 0 && t->is_cuda())
            c10::hip::HIPCachingAllocatorMasqueradingAsCUDA::recordStreamMasqueradingAsCUDA(t->storage().data_ptr(), s);
}

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

// ---- several tensors over ONE allocation.  A forward used to make twelve at::empty calls and a backward three plus
// sixteen narrow / view calls for the gradient arena's segments -- each a trip through the dispatcher and, for the
// allocations, the caching allocator's lock.  view_of builds the TensorImpl of a contiguous window into `base`'s storage
// directly (what as_strided does underneath, without the dispatch); byte offsets are multiples of 256.
inline size_t up256(size_t n) { return (n + 255) & ~(size_t)255; }
// The raw construction below was validated on PyTorch 2.10 (TensorImpl::VIEW constructor, set_sizes_contiguous, set_storage_offset,
// wrap_tensor_impl; the autograd engine's saved-tensor version check on such outputs: tests/test_hip_binding_guard.py).  Built
// against another PyTorch it is NOT used unless DGR_RAW_VIEWS=1 asks for it; DGR_RAW_VIEWS=0 switches it off anywhere; and the
// first view made in a process is checked against the dispatcher's own view of the same window (pointer, sizes, strides, dtype,
// aliasing, a fresh version counter) -- a mismatch falls back for good, with one warning.  The fall-back is at::from_blob over the
// window, its deleter holding `base`: a tensor of its own (own storage object, own version counter, not a view in autograd's
// books), like the raw one.  NOT narrow / view / as_strided: a custom Function that returns several views of one base may not have
// them edited in place at all, and views of one base share a version counter -- editing `color` would then invalidate the saved
// `opacity_map`; neither happens with the reference's separately allocated outputs.
inline Tensor dispatcher_view(const Tensor& base, size_t byte_off, c10::IntArrayRef sizes, at::ScalarType dt) {
    Tensor keep = base;
    return at::from_blob(static_cast<char*>(base.data_ptr()) + byte_off, sizes, [keep](void*) mutable { keep = Tensor(); },
                         at::TensorOptions().dtype(dt).device(base.device()));
}
inline Tensor raw_view(const Tensor& base, size_t byte_off, c10::IntArrayRef sizes, at::ScalarType dt) {
    auto impl = c10::make_intrusive<c10::TensorImpl>(c10::TensorImpl::VIEW, c10::Storage(base.storage()), base.key_set(),
                                                     c10::scalarTypeToTypeMeta(dt));
    impl->set_sizes_contiguous(sizes);
    impl->set_storage_offset((int64_t)(byte_off / c10::elementSize(dt)));
    return Tensor::wrap_tensor_impl(std::move(impl));
}
std::atomic<int> g_raw_views{-1};  // -1: not decided yet, 0: dispatcher views, 1: raw views
inline bool decide_raw_views(const Tensor& base, size_t byte_off, c10::IntArrayRef sizes, at::ScalarType dt) {
    const char* e = getenv("DGR_RAW_VIEWS");
    if (e && e[0] == '0') return false;
    // validated on PyTorch 2.10; later releases take the self-check below (which the first view of every process runs anyway),
    // earlier ones the dispatcher's windows unless DGR_RAW_VIEWS=1 asks for the check
#if !(defined(TORCH_VERSION_MAJOR) && (TORCH_VERSION_MAJOR > 2 || (TORCH_VERSION_MAJOR == 2 && TORCH_VERSION_MINOR >= 10)))
    if (!(e && e[0] == '1')) return false;
#endif
    bool ok = false;
    try {
        const Tensor a = raw_view(base, byte_off, sizes, dt), b = dispatcher_view(base, byte_off, sizes, dt);
        ok = a.data_ptr() == b.data_ptr() && a.sizes() == b.sizes() && a.strides() == b.strides() && a.scalar_type() == b.scalar_type() &&
             a.device() == b.device() && a.is_alias_of(base) && a._version() == 0 && a.is_contiguous() && !a.requires_grad() &&
             !a.is_view() && a.numel() == b.numel() && a.key_set() == b.key_set();
    } catch (...) {
        ok = false;
    }
    if (!ok) TORCH_WARN_ONCE("dgr_hip: the raw tensor views of csrc/torch_ext.cpp do not behave as on the PyTorch they were validated on; "
                             "using at::from_blob windows instead");
    return ok;
}
inline Tensor view_of(const Tensor& base, size_t byte_off, c10::IntArrayRef sizes, at::ScalarType dt) {
    int mode = g_raw_views.load(std::memory_order_relaxed);
    if (mode < 0) {
        mode = decide_raw_views(base, byte_off, sizes, dt) ? 1 : 0;
        g_raw_views.store(mode, std::memory_order_relaxed);
    }
    return mode ? raw_view(base, byte_off, sizes, dt) : dispatcher_view(base, byte_off, sizes, dt);
}
inline Tensor bytes_on(const c10::Device& dev, size_t n) {
    return at::empty({(long long)std::max<size_t>(n, 1)}, at::TensorOptions().dtype(at::kByte).device(dev));
}

// The three opaque state buffers of a presized forward as windows of one allocation (they are saved and released together).
struct StateArena {
    Tensor geom, binning, img;
    StateArena(const c10::Device& dev, int P, int W, int H, long cap) {
        const size_t ng = up256(dgr_geometry_bytes(P)), ni = up256(dgr_image_bytes(W, H)), nb = up256(dgr_binning_bytes((int)cap, W, H));
        const Tensor a = bytes_on(dev, ng + ni + nb);
        geom = view_of(a, 0, {(long long)ng}, at::kByte);
        img = view_of(a, ng, {(long long)ni}, at::kByte);
        binning = view_of(a, ng + ni, {(long long)nb}, at::kByte);
    }
};

// Strict mode's one host wait: the forward reports through an armed status slot (pinned host memory written by the forward
// blend's first workgroup, include/dgr_hip.h: dgr_status_arm) and the host polls that memory -- the reference's blocking copy
// of num_rendered (L/cuda_rasterizer/rasterizer_impl.cu:287) without the copy, the event and the wake-up, and with the
// longest-list report that lets the next forward of the shape skip the tile schedule.  While a hipGraph is recorded nothing
// can be read back: strict mode cannot be captured (as before).
template <typename Run>
inline void strict_status(Run& run, long cap, int* s, void* st) {
    // (a status word cannot be read back while the stream records a hipGraph: the wait below would never end)
    if (dgr_stream_is_capturing(st))
        throw std::runtime_error("strict status mode (one host wait per forward) cannot run while its stream is being captured into a "
                                 "hipGraph: use the lazy mode (DGR_SYNC_MODE=lazy), after a few eager forwards of the same shape");
    const long ticket = dgr_status_arm();
    check(ticket);
    try {
        run(cap);
    } catch (...) {
        int unused[4];
        (void)dgr_status_poll(ticket, 1, unused);  // (completed by the library: releases the slot)
        throw;
    }
    check(dgr_status_poll(ticket, 1, s));
}

// mode: 0 = callback entry point (the strict mirror: allocation callbacks + the reference's blocking read),
//       1 = presized, strict: one host wait until num_rendered is known; retries a too-small capacity itself,
//       2 = presized, lazy: no host synchronisation; returns a status ticket (dgr_status_post) or -1 while capturing.
struct LightFwd {
    long rendered = -1, ticket = -1, cap = 0;
    Tensor status, color, depth, median, var, alpha, radii, geom, binning, img, unc, px;
};
LightFwd light_forward_core(const Tensor& background, const Tensor& means3D_, const Tensor& colors_, const Tensor& opacity_,
                            const Tensor& scales_, const Tensor& rotations_, double scale_modifier, const Tensor& cov3D_,
                            const Tensor& viewmatrix_, const Tensor& gt_depth_, const Tensor& projmatrix_, double tan_fovx,
                            double tan_fovy, long H, long W, const Tensor& sh_, long degree, const Tensor& campos_,
                            bool prefiltered, bool debug, long capacity, long mode) {
    if (means3D_.dim() != 2 || means3D_.size(1) != 3) throw std::runtime_error("means3D must have dimensions (num_points, 3)");
    const c10::Device dev = means3D_.device();
    if (!dev.is_cuda()) throw std::runtime_error("dgr_hip runs on the GPU only (no CPU path exists, as in the reference)");
    Probe p_pre(HP_PRELUDE);
    c10::hip::HIPGuardMasqueradingAsCUDA guard(dev);
    const int P = (int)means3D_.size(0);
    const Tensor means3D = f32c(means3D_, dev), bg = f32c(background, dev), colors = f32c(colors_, dev),
                 opacity = f32c(opacity_, dev), scales = f32c(scales_, dev), rotations = f32c(rotations_, dev),
                 cov3D = f32c(cov3D_, dev), view = f32c(viewmatrix_, dev), proj = f32c(projmatrix_, dev),
                 campos = f32c(campos_, dev), gt = f32c(gt_depth_, dev), sh = f32c(sh_, dev);
    const int M = sh.numel() != 0 ? (int)sh.size(1) : 0;
    keep_until_read(dev, {&means3D_, &background, &colors_, &opacity_, &scales_, &rotations_, &cov3D_, &viewmatrix_, &projmatrix_,
                          &campos_, &gt_depth_, &sh_});
    const auto u8 = at::TensorOptions().dtype(at::kByte).device(dev);
    p_pre.stop();
    Probe p_out(HP_OUT_ALLOC);
    LightFwd o;
    // allocation 1: the five images (every pixel is written by the blend kernel); allocation 2: radii (written for every
    // Gaussian), the two median statistics (cleared by the kernels) and the status word
    const size_t N = (size_t)H * (size_t)W, n1 = up256(4 * N), np = up256(4 * (size_t)P);
    const Tensor images = bytes_on(dev, up256(12 * N) + 4 * n1);
    o.color = view_of(images, 0, {3, H, W}, at::kFloat);
    o.depth = view_of(images, up256(12 * N), {1, H, W}, at::kFloat);
    o.median = view_of(images, up256(12 * N) + n1, {1, H, W}, at::kFloat);
    o.var = view_of(images, up256(12 * N) + 2 * n1, {1, H, W}, at::kFloat);
    o.alpha = view_of(images, up256(12 * N) + 3 * n1, {1, H, W}, at::kFloat);
    const Tensor per_gaussian = bytes_on(dev, 3 * np + 256);
    o.radii = view_of(per_gaussian, 0, {P}, at::kInt);
    o.unc = view_of(per_gaussian, np, {P, 1}, at::kFloat);
    o.px = view_of(per_gaussian, 2 * np, {P, 1}, at::kInt);
    o.status = view_of(per_gaussian, 3 * np, {4}, at::kInt);
    void* st = stream_of(dev);
    p_out.stop();

    if (mode == 0 || P == 0) {
        o.geom = at::empty({0}, u8); o.binning = at::empty({0}, u8); o.img = at::empty({0}, u8);
        Alloc3 al{{&o.geom, dev}, {&o.binning, dev}, {&o.img, dev}};
        const int rc = dgr_light_forward(st, cb_geom, cb_binning, cb_img, &al, P, (int)degree, M, ptr<float>(bg), (int)W, (int)H,
                                         ptr<float>(means3D), ptr<float>(sh), ptr<float>(colors), ptr<float>(opacity),
                                         ptr<float>(scales), (float)scale_modifier, ptr<float>(rotations), ptr<float>(cov3D),
                                         ptr<float>(view), ptr<float>(proj), ptr<float>(campos), (float)tan_fovx,
                                         (float)tan_fovy, prefiltered ? 1 : 0, ptr<float>(o.color), ptr<float>(o.depth),
                                         ptr<float>(o.median), ptr<float>(o.alpha), ptr<float>(gt), ptr<float>(o.var),
                                         ptr<float>(o.unc), ptr<int>(o.px), ptr<int>(o.radii), debug ? 1 : 0);
        check(rc);
        o.rendered = o.cap = rc;
        return o;
    }
    auto run = [&](long cap) {
        Probe p_st(HP_STATE_ALLOC);
        const StateArena sa(dev, P, (int)W, (int)H, cap);  // allocation 3
        o.geom = sa.geom; o.binning = sa.binning; o.img = sa.img;
        p_st.stop();
        Probe p_c(HP_FWD_C);
        check(dgr_light_forward_presized(st, (char*)o.geom.data_ptr(), (char*)o.binning.data_ptr(), (int)cap, (char*)o.img.data_ptr(),
                                         o.status.data_ptr<int>(), P, (int)degree, M, ptr<float>(bg), (int)W, (int)H,
                                         ptr<float>(means3D), ptr<float>(sh), ptr<float>(colors), ptr<float>(opacity),
                                         ptr<float>(scales), (float)scale_modifier, ptr<float>(rotations), ptr<float>(cov3D),
                                         ptr<float>(view), ptr<float>(proj), ptr<float>(campos), (float)tan_fovx,
                                         (float)tan_fovy, prefiltered ? 1 : 0, ptr<float>(o.color), ptr<float>(o.depth),
                                         ptr<float>(o.median), ptr<float>(o.alpha), ptr<float>(gt), ptr<float>(o.var),
                                         ptr<float>(o.unc), ptr<int>(o.px), ptr<int>(o.radii)));
    };
    if (mode == 2) {
        // the status word comes back through pinned host memory written by the binning kernel (dgr_status_arm): no copy, no
        // event; while a hipGraph is being recorded nothing can be read back
        {
            Probe p_arm(HP_ARM);
            if (!dgr_stream_is_capturing(st)) {
                o.ticket = dgr_status_arm();
                check(o.ticket);
            }
        }
        try {
            run(capacity);
        } catch (...) {
            int unused[4];
            if (o.ticket >= 0) (void)dgr_status_poll(o.ticket, 1, unused);  // (completed by the library: releases the slot)
            throw;
        }
        o.cap = capacity;
        return o;
    }
    long cap = capacity;
    for (;;) {
        int s[4] = {0, 0, 0, 0};
        strict_status(run, cap, s, st);  // the one host wait of this forward: until num_rendered is known
        if (s[2]) throw std::runtime_error("Point is filtered although prefiltered is set. This shouldn't happen!");
        o.rendered = s[0];
        if (o.rendered <= cap) break;
        cap = (long)(o.rendered * 1.1) + 4096;  // overflow: every tile list was left empty; run again
    }
    o.cap = cap;
    if (debug) check(hipStreamSynchronize((hipStream_t)st) == hipSuccess ? 0 : DGR_ERR_HIP);
    return o;
}

// The `_C.rasterize_gaussians` of the light variant (L/rasterize_points.cu:35-129) plus the policy values the Python side
// keeps.  Returns (num_rendered or -1, ticket or -1, capacity used, device status word, color, depth, median, var, alpha,
// radii, geom, binning, img, gau_uncertainty, gau_related_pixels).
std::tuple<long, long, long, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor>
light_forward(const Tensor& background, const Tensor& means3D, const Tensor& colors, const Tensor& opacity,
              const Tensor& scales, const Tensor& rotations, double scale_modifier, const Tensor& cov3D,
              const Tensor& viewmatrix, const Tensor& gt_depth, const Tensor& projmatrix, double tan_fovx,
              double tan_fovy, long H, long W, const Tensor& sh, long degree, const Tensor& campos, bool prefiltered,
              bool debug, long capacity, long mode) {
    const LightFwd o = light_forward_core(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D, viewmatrix,
                                          gt_depth, projmatrix, tan_fovx, tan_fovy, H, W, sh, degree, campos, prefiltered, debug,
                                          capacity, mode);
    return {o.rendered, o.ticket, o.cap, o.status, o.color, o.depth, o.median, o.var, o.alpha, o.radii, o.geom, o.binning, o.img, o.unc, o.px};
}

// ---- resident backward scratch.  The backward's accumulator rows (64 bytes per Gaussian) must be zero when the blend
// backward starts; a fresh allocation per call needs a clearing launch in front of it.  Instead one buffer per (device,
// stream, size) is kept across calls: zero-filled when created, and every backward leaves it zero again (include/dgr_hip.h:
// dgr_backward_scratch_clean_arm -- the per-Gaussian kernel clears the rows it reads).  Calls that share a buffer run on one
// stream, i.e. in order.  Not while a hipGraph is being recorded (a replay may run on any stream, next to anything): a
// capture gets a fresh buffer and the clearing launch.  DGR_RESIDENT_SCRATCH=0 switches the cache off.
struct ScratchEntry {
    int device;
    void* stream;
    size_t bytes;
    Tensor buf;
    uint64_t stamp;
};
std::mutex g_scr_mu;
std::vector<ScratchEntry>& scratch_cache() {
    static auto* v = new std::vector<ScratchEntry>();  // never destroyed: tensors must not outlive the allocator at exit
    return *v;
}
uint64_t g_scr_clock = 0;
const bool g_resident_scratch = [] { const char* e = getenv("DGR_RESIDENT_SCRATCH"); return !(e && e[0] == '0'); }();
Tensor backward_scratch(const c10::Device& dev, void* stream, size_t nbytes, bool* resident) {
    *resident = false;
    if (!g_resident_scratch || dgr_stream_is_capturing(stream)) return bytes_on(dev, nbytes);
    std::lock_guard<std::mutex> lk(g_scr_mu);
    auto& c = scratch_cache();
    for (auto& e : c)
        if (e.device == dev.index() && e.stream == stream && e.bytes == nbytes) {
            e.stamp = ++g_scr_clock;
            *resident = true;
            return e.buf;
        }
    if (c.size() >= 32) {  // drop the entry used longest ago (its memory goes back to the caching allocator)
        size_t old = 0;
        for (size_t i = 1; i < c.size(); i++)
            if (c[i].stamp < c[old].stamp) old = i;
        c.erase(c.begin() + (long)old);
    }
    Tensor buf = at::zeros({(long long)std::max<size_t>(nbytes, 1)}, at::TensorOptions().dtype(at::kByte).device(dev));
    c.push_back(ScratchEntry{dev.index(), stream, nbytes, buf, ++g_scr_clock});
    *resident = true;
    return buf;
}
void drop_scratch(const c10::Device& dev, void* stream) {  // after a failed call the buffer's contents are unknown
    std::lock_guard<std::mutex> lk(g_scr_mu);
    auto& c = scratch_cache();
    for (size_t i = 0; i < c.size();)
        if (c[i].device == dev.index() && c[i].stream == stream) c.erase(c.begin() + (long)i); else i++;
}

// The flat arena of the eight per-Gaussian gradients (dgr_amd.light._grad_arena): segments in the order means3D, means2D,
// sh, opacity, scales, rotations | cov3D, colors, 256-byte aligned; the first six are one contiguous span = the multi-GPU
// all-reduce payload.  g[] receives them in the return order of the reference binding: means2D, colors, opacity, means3D,
// cov3D, sh, scales, rotations.  Every row is written by the kernels (zeros for invisible Gaussians): no zero-fill.
inline void grad_arena(const c10::Device& dev, int P, int M, Tensor* g) {
    const long long n[8] = {3LL * P, 3LL * P, 3LL * M * P, P, 3LL * P, 4LL * P, 6LL * P, 3LL * P};
    size_t off[8], o = 0;
    for (int i = 0; i < 8; i++) { off[i] = o; o += up256(4 * (size_t)n[i]); }
    const Tensor arena = at::empty({(long long)std::max<size_t>(o / 4, 1)}, at::TensorOptions().dtype(at::kFloat).device(dev));
    g[3] = view_of(arena, off[0], {P, 3}, at::kFloat); g[0] = view_of(arena, off[1], {P, 3}, at::kFloat);
    g[5] = view_of(arena, off[2], {P, M, 3}, at::kFloat); g[2] = view_of(arena, off[3], {P, 1}, at::kFloat);
    g[6] = view_of(arena, off[4], {P, 3}, at::kFloat); g[7] = view_of(arena, off[5], {P, 4}, at::kFloat);
    g[4] = view_of(arena, off[6], {P, 6}, at::kFloat); g[1] = view_of(arena, off[7], {P, 3}, at::kFloat);
}

// L/rasterize_points.cu:131-236.  Returns (dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales,
// dL_drotations, dL_dview [1,4,4]); the first eight are windows of one flat arena (grad_arena above), or undefined tensors
// (None) when need_gaussian_grads is false (tracking: the library then skips every dense per-Gaussian row).
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
                 gt = f32c(gt_depth_, dev), sh = f32c(sh_, dev), alphas = f32c(alphas_, dev), perspec = f32c_diag4(perspec_, dev),
                 gC = f32c(dL_dout_color, dev), gD = f32c(dL_dout_depth, dev), gM = f32c(dL_dout_median, dev),
                 gV = f32c(dL_dout_var, dev);
    const int M = sh.numel() != 0 ? (int)sh.size(1) : 0;
    // (the forward recorded the saved inputs; the gradient images of a graph root are whatever the caller passed in)
    keep_until_read(dev, {&dL_dout_color, &dL_dout_depth, &dL_dout_median, &dL_dout_var, &alphas_, &perspec_});
    Probe p_al(HP_BWD_ALLOC);
    std::vector<Tensor> g(9);
    float* gp[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    if (need_gaussian_grads) {
        grad_arena(dev, P, M, g.data());
        for (int i = 0; i < 8; i++) gp[i] = ptr<float>(g[i]);
    } else {
        map_off = true;  // nobody reads the per-Gaussian sums: the blend kernel forms the three pose sums only
    }
    // scratch (accumulator rows, cleared by the backward's first launch) and, behind it, the [1,4,4] pose gradient -- what
    // L/__init__.py:160-161 sums over dim 0
    // (with the option "deterministic_grads" the scratch also holds 64 bytes per tile instance: R = num_rendered, or a lazy forward's capacity)
    const size_t nscr = up256(dgr_light_backward_scratch_bytes_r(P, (int)W, (int)H, (int)R));
    void* st = stream_of(dev);
    bool resident = false;
    const Tensor scratch = backward_scratch(dev, st, nscr, &resident);
    Tensor dview = at::empty({1, 4, 4}, at::TensorOptions().dtype(at::kFloat).device(dev));
    if (resident) dgr_backward_scratch_clean_arm();
    p_al.stop();
    Probe p_bc(HP_BWD_C);
    const int rc = (dgr_light_backward(st, P, (int)degree, M, (int)R, ptr<float>(bg), (int)W, (int)H, ptr<float>(means3D),
                             ptr<float>(sh), ptr<float>(colors), ptr<float>(alphas), ptr<float>(scales), (float)scale_modifier,
                             ptr<float>(rotations), ptr<float>(cov3D), ptr<float>(view), ptr<float>(proj), ptr<float>(campos),
                             (float)tan_fovx, (float)tan_fovy, ptr<int>(radii), bytes(geomBuffer),
                             bytes(binningBuffer), bytes(imageBuffer), ptr<float>(gC),
                             ptr<float>(gD), ptr<float>(gM), ptr<float>(gV), gp[0], nullptr, gp[2], gp[1], nullptr, gp[3], gp[4],
                             gp[5], gp[6], gp[7], debug ? 1 : 0, nullptr, ptr<float>(perspec), dview.data_ptr<float>(), nullptr,
                             ptr<float>(gt), track_off ? 1 : 0, map_off ? 1 : 0, (char*)scratch.data_ptr(), nscr));
    if (rc < 0 && resident) drop_scratch(dev, st);
    check(rc);
    g[8] = std::move(dview);
    return g;
}


// ------------------------------------------------------------------------------------------------ full variant
// F/rasterize_points.cu:35-120.  Modes as light_forward_core.
struct FullFwd {
    long rendered = -1, related = -1, ticket = -1, cap = 0;
    Tensor status, color, depth, unc, radii, geom, binning, img;
};
FullFwd full_forward_core(const Tensor& background, const Tensor& means3D_, const Tensor& colors_, const Tensor& opacity_,
                          const Tensor& scales_, const Tensor& rotations_, double scale_modifier, const Tensor& cov3D_,
                          const Tensor& viewmatrix_, const Tensor& gt_depth_, const Tensor& projmatrix_, double tan_fovx,
                          double tan_fovy, long H, long W, const Tensor& sh_, long degree, const Tensor& campos_,
                          bool prefiltered, long capacity, long mode, bool want_related = true) {
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
    keep_until_read(dev, {&means3D_, &background, &colors_, &opacity_, &scales_, &rotations_, &cov3D_, &viewmatrix_, &projmatrix_,
                          &campos_, &gt_depth_, &sh_});
    const auto u8 = at::TensorOptions().dtype(at::kByte).device(dev);
    FullFwd o;
    const size_t N = (size_t)H * (size_t)W, n1 = up256(4 * N), np = up256(4 * (size_t)P);
    const Tensor images = bytes_on(dev, up256(12 * N) + 2 * n1);
    o.color = view_of(images, 0, {3, H, W}, at::kFloat);
    o.depth = view_of(images, up256(12 * N), {1, H, W}, at::kFloat);
    o.unc = view_of(images, up256(12 * N) + n1, {1, H, W}, at::kFloat);
    const Tensor per_gaussian = bytes_on(dev, np + 256);
    o.radii = view_of(per_gaussian, 0, {P}, at::kInt);  // (written for every Gaussian by preprocess_fwd)
    o.status = view_of(per_gaussian, np, {4}, at::kInt);
    void* st = stream_of(dev);
    if (mode == 0 || P == 0) {
        o.geom = at::empty({0}, u8); o.binning = at::empty({0}, u8); o.img = at::empty({0}, u8);
        Alloc3 al{{&o.geom, dev}, {&o.binning, dev}, {&o.img, dev}};
        int ng = 0;
        const int rc = dgr_full_forward(st, cb_geom, cb_binning, cb_img, &al, P, (int)degree, M, ptr<float>(bg), (int)W, (int)H,
                                        ptr<float>(means3D), ptr<float>(sh), ptr<float>(colors), ptr<float>(opacity),
                                        ptr<float>(scales), (float)scale_modifier, ptr<float>(rotations), ptr<float>(cov3D),
                                        ptr<float>(view), ptr<float>(proj), ptr<float>(campos), (float)tan_fovx, (float)tan_fovy,
                                        prefiltered ? 1 : 0, ptr<float>(o.color), ptr<float>(o.depth), ptr<float>(gt), ptr<float>(o.unc),
                                        ptr<int>(o.radii), &ng);
        check(rc);
        o.rendered = o.cap = rc;
        o.related = ng;
        return o;
    }
    auto run = [&](long cap) {
        const StateArena sa(dev, P, (int)W, (int)H, cap);
        o.geom = sa.geom; o.binning = sa.binning; o.img = sa.img;
        check(dgr_full_forward_presized(st, (char*)o.geom.data_ptr(), (char*)o.binning.data_ptr(), (int)cap, (char*)o.img.data_ptr(),
                                        o.status.data_ptr<int>(), P, (int)degree, M, ptr<float>(bg), (int)W, (int)H,
                                        ptr<float>(means3D), ptr<float>(sh), ptr<float>(colors), ptr<float>(opacity),
                                        ptr<float>(scales), (float)scale_modifier, ptr<float>(rotations), ptr<float>(cov3D),
                                        ptr<float>(view), ptr<float>(proj), ptr<float>(campos), (float)tan_fovx, (float)tan_fovy,
                                        prefiltered ? 1 : 0, ptr<float>(o.color), ptr<float>(o.depth), ptr<float>(gt), ptr<float>(o.unc),
                                        ptr<int>(o.radii)));
    };
    if (mode == 2) {
        // the status word comes back through pinned host memory written by the binning kernel (dgr_status_arm): no copy, no
        // event; while a hipGraph is being recorded nothing can be read back
        {
            Probe p_arm(HP_ARM);
            if (!dgr_stream_is_capturing(st)) {
                o.ticket = dgr_status_arm();
                check(o.ticket);
            }
        }
        try {
            run(capacity);
        } catch (...) {
            int unused[4];
            if (o.ticket >= 0) (void)dgr_status_poll(o.ticket, 1, unused);  // (completed by the library: releases the slot)
            throw;
        }
        o.cap = capacity;
        return o;
    }
    long cap = capacity;
    for (;;) {
        int s[4] = {0, 0, 0, 0};
        strict_status(run, cap, s, st);
        if (s[2]) throw std::runtime_error("Point is filtered although prefiltered is set. This shouldn't happen!");
        o.rendered = s[0];
        if (o.rendered <= cap) break;
        cap = (long)(o.rendered * 1.1) + 4096;
    }
    o.cap = cap;
    // num_related (the reference's NG) is produced by the forward blend: the reference's second blocking read
    // (F/cuda_rasterizer/rasterizer_impl.cu:498) -- a wait for the whole forward.  The `_C.rasterize_gaussians` mirror returns the
    // number, as the reference's does; the autograd node does not wait for it: NG only sizes the reference's pair lists in ITS
    // backward (F/__init__.py:91,100,130), which this backward does not have (csrc/render_full.hip), and no caller of
    // GaussianRasterizer.forward ever sees it.  The strict forward then returns as soon as num_rendered is known, the blend
    // still running (config 2, one view at a time in the default mode: 0.205 -> 0.17 ms).
    if (want_related) o.related = o.status.to(at::kCPU).data_ptr<int>()[3];
    return o;
}

// Returns (num_rendered or -1, num_related or -1, ticket or -1, capacity used, device status word, color, depth,
// uncertainty, radii, geom, binning, img).
std::tuple<long, long, long, long, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor>
full_forward(const Tensor& background, const Tensor& means3D, const Tensor& colors, const Tensor& opacity, const Tensor& scales,
             const Tensor& rotations, double scale_modifier, const Tensor& cov3D, const Tensor& viewmatrix,
             const Tensor& gt_depth, const Tensor& projmatrix, double tan_fovx, double tan_fovy, long H, long W,
             const Tensor& sh, long degree, const Tensor& campos, bool prefiltered, long capacity, long mode) {
    const FullFwd o = full_forward_core(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D, viewmatrix,
                                        gt_depth, projmatrix, tan_fovx, tan_fovy, H, W, sh, degree, campos, prefiltered, capacity, mode);
    return {o.rendered, o.related, o.ticket, o.cap, o.status, o.color, o.depth, o.unc, o.radii, o.geom, o.binning, o.img};
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
                 gt = f32c(gt_depth_, dev), sh = f32c(sh_, dev), perspec = f32c_diag4(perspec_, dev), gC = f32c(dL_dout_color, dev),
                 gD = f32c(dL_dout_depth, dev), gU = f32c(dL_dout_unc, dev);
    const int M = sh.numel() != 0 ? (int)sh.size(1) : 0;
    keep_until_read(dev, {&dL_dout_color, &dL_dout_depth, &dL_dout_unc, &perspec_});
    std::vector<Tensor> g(9);
    float* gp[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    if (need_gaussian_grads) {
        grad_arena(dev, P, M, g.data());
        for (int i = 0; i < 8; i++) gp[i] = ptr<float>(g[i]);
    }
    Tensor dview = at::empty({4, 4}, at::TensorOptions().dtype(at::kFloat).device(dev));
    const size_t nscr = up256(dgr_light_backward_scratch_bytes_r(P, (int)W, (int)H, (int)R));  // (deterministic_grads: + rows per instance)
    void* st = stream_of(dev);
    bool resident = false;
    const Tensor scratch = backward_scratch(dev, st, nscr, &resident);
    if (resident) dgr_backward_scratch_clean_arm();
    // gp: [0] means2D [1] colors [2] opacity [3] means3D [4] cov3D [5] sh [6] scales [7] rotations
    const int rc = (dgr_full_backward(st, P, (int)degree, M, (int)R, ptr<float>(bg), (int)W, (int)H, ptr<float>(means3D),
                            ptr<float>(sh), ptr<float>(colors), ptr<float>(scales), (float)scale_modifier, ptr<float>(rotations),
                            ptr<float>(cov3D), ptr<float>(view), ptr<float>(proj), ptr<float>(campos), (float)tan_fovx,
                            (float)tan_fovy, ptr<int>(radii), bytes(geomBuffer), bytes(binningBuffer), bytes(imageBuffer),
                            ptr<float>(gC), ptr<float>(gD), gp[0], nullptr, gp[2], gp[1], gp[3], gp[4], gp[5], gp[6], gp[7], nullptr,
                            nullptr, nullptr, nullptr, nullptr, ptr<float>(perspec), nullptr, nullptr, nullptr,
                            dview.data_ptr<float>(), nullptr, nullptr, nullptr, ptr<float>(gt), ptr<float>(gU),
                            (char*)scratch.data_ptr(), nscr));
    if (rc < 0 && resident) drop_scratch(dev, st);
    check(rc);
    g[8] = std::move(dview);
    return g;
}

// ------------------------------------------------------------------------------------------------ autograd nodes
// The reference's surface is an eager autograd.Function written in Python over `_C` (L/__init__.py:46-176,
// F/__init__.py:46-151).  Kept as that (dgr_amd.light / full._RasterizeGaussians: the debug path, the ctypes binding), it
// costs a Python frame, a tuple of forty arguments and a context object per forward, and in the backward a hand-off from the
// autograd engine's thread into the interpreter -- together more than the GPU needs for a 640x480 / 100 k view (BASELINE
// config 2).  The nodes below are the same Function in C++: ONE Python -> C++ crossing per forward, and a backward that runs
// inside the engine without the interpreter.  Same inputs in the same order, same saved state, same outputs, same None
// gradients for gt_depth and the settings.
//
// dgr_amd.multiview.ViewStreams.before_backward(): an event the backward THAT RUNS ON A GIVEN STREAM makes that stream wait
// for once its kernels are issued -- before autograd goes on to the accumulation into shared `.grad`.  Keyed by the raw
// stream handle; the Python side keeps the torch objects alive until it drops the entry.
std::mutex g_wait_mu;
std::vector<std::pair<void*, void*>> g_post_backward_waits;  // (stream, event)
void set_post_backward_wait(long stream, long event) {
    std::lock_guard<std::mutex> lk(g_wait_mu);
    for (auto& e : g_post_backward_waits)
        if (e.first == (void*)stream) { e.second = (void*)event; return; }
    g_post_backward_waits.emplace_back((void*)stream, (void*)event);
}
void drop_post_backward_wait(long stream) {
    std::lock_guard<std::mutex> lk(g_wait_mu);
    for (size_t i = 0; i < g_post_backward_waits.size(); i++)
        if (g_post_backward_waits[i].first == (void*)stream) { g_post_backward_waits.erase(g_post_backward_waits.begin() + (long)i); return; }
}
void consume_post_backward_wait(void* stream) {
    void* ev = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_wait_mu);
        for (size_t i = 0; i < g_post_backward_waits.size(); i++)
            if (g_post_backward_waits[i].first == stream) {
                ev = g_post_backward_waits[i].second;
                g_post_backward_waits.erase(g_post_backward_waits.begin() + (long)i);
                break;
            }
    }
    if (ev) check(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)ev, 0) == hipSuccess ? 0 : DGR_ERR_HIP);
}

using torch::autograd::AutogradContext;
using torch::autograd::variable_list;

// what a forward reports beside its output tensors (read by light_apply / full_apply right after Function::apply)
struct FwdReport {
    long rendered = -1, related = -1, ticket = -1, cap = 0;
    Tensor status;
};
thread_local FwdReport g_report;

inline Tensor zeros_like_image(long c, long H, long W, const c10::Device& dev) {
    return at::zeros({c, H, W}, at::TensorOptions().dtype(at::kFloat).device(dev));
}

// a block under a word of dgr_thread_options_effective() (include/dgr_hip.h): a backward under its forward's options
struct UnderOptions {
    int prev;
    explicit UnderOptions(int word) : prev(dgr_thread_options_swap(word)) {}
    ~UnderOptions() { dgr_thread_options_swap(prev); }
};

struct LightNode : public torch::autograd::Function<LightNode> {
    // inputs 0..9 as L/__init__.py:46-60; then the settings' tensors and scalars (L/__init__.py:180-195) and the binning policy
    static variable_list forward(AutogradContext* ctx, const Tensor& means3D, const Tensor& means2D, const Tensor& sh,
                                 const Tensor& colors_precomp, const Tensor& opacities, const Tensor& scales, const Tensor& rotations,
                                 const Tensor& cov3Ds_precomp, const Tensor& viewmatrix, const Tensor& gt_depth, const Tensor& bg,
                                 const Tensor& projmatrix, const Tensor& campos, const Tensor& perspec, double scale_modifier,
                                 double tanfovx, double tanfovy, int64_t H, int64_t W, int64_t degree, bool prefiltered,
                                 bool track_off, bool map_off, int64_t capacity, int64_t mode) {
        (void)means2D;  // never read (L/__init__.py:66-87): it exists so that autograd hands dL_dmeans2D back
        Probe p_f(HP_FWD);
        LightFwd o = light_forward_core(bg, means3D, colors_precomp, opacities, scales, rotations, scale_modifier, cov3Ds_precomp,
                                        viewmatrix, gt_depth, projmatrix, tanfovx, tanfovy, H, W, sh, degree, campos, prefiltered,
                                        false, capacity, mode);
        g_report.rendered = o.rendered; g_report.ticket = o.ticket; g_report.cap = o.cap; g_report.status = o.status;
        Probe p_s(HP_SAVE);
        // L/__init__.py:101-102 (+ the settings' tensors, which the Python Function reads from ctx.raster_settings)
        ctx->save_for_backward({colors_precomp, means3D, scales, rotations, cov3Ds_precomp, viewmatrix, o.radii, sh, o.geom,
                                o.binning, o.img, o.alpha, gt_depth, bg, projmatrix, campos, perspec});
        auto& d = ctx->saved_data;
        d["scale_modifier"] = scale_modifier; d["tanfovx"] = tanfovx; d["tanfovy"] = tanfovy; d["degree"] = degree;
        d["R"] = (int64_t)(o.rendered >= 0 ? o.rendered : o.cap); d["track_off"] = track_off; d["map_off"] = map_off;
        d["H"] = H; d["W"] = W;
        d["options"] = (int64_t)dgr_thread_options_effective();  // the backward runs under the forward's per-call options
        // four of the eight outputs (radii, opacity_map, gau_uncertainty, gau_related_pixels) have no gradient input in the
        // backward: no zero-filled gradient tensors for them
        ctx->set_materialize_grads(false);
        ctx->mark_non_differentiable({o.radii, o.px});
        return {o.color, o.radii, o.depth, o.median, o.var, o.alpha, o.unc, o.px};
    }

    static variable_list backward(AutogradContext* ctx, variable_list grad) {
        Probe p_b(HP_BWD);
        Probe p_u(HP_UNPACK);
        const auto sv = ctx->get_saved_variables();
        auto& d = ctx->saved_data;
        const Tensor& means3D = sv[1];
        const c10::Device dev = means3D.device();
        const long H = d["H"].toInt(), W = d["W"].toInt();
        p_u.stop();
        // an output that did not take part in the loss arrives undefined: zeros, as the reference's autograd would have passed
        const Tensor gC = grad[0].defined() ? grad[0] : zeros_like_image(3, H, W, dev);
        const Tensor gD = grad[2].defined() ? grad[2] : zeros_like_image(1, H, W, dev);
        // (no gradient image for the median depth / the depth variance: NULL at the C ABI, which then runs the lean blend backward
        //  -- no zero images are made, filled and read)
        const Tensor gM = grad[3].defined() ? grad[3] : Tensor();
        const Tensor gV = grad[4].defined() ? grad[4] : Tensor();
        bool need = false;  // (means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp): tracking needs none
        for (int i = 0; i < 8; i++) need = need || ctx->needs_input_grad(i);
        const UnderOptions under((int)d["options"].toInt());  // (the engine may run this node on a thread of its own)
        std::vector<Tensor> g = light_backward(sv[13], means3D, sv[6], sv[0], sv[2], sv[3], d["scale_modifier"].toDouble(), sv[4],
                                               sv[5], sv[14], d["tanfovx"].toDouble(), d["tanfovy"].toDouble(), gC, gD, gM, gV, sv[12],
                                               sv[7], d["degree"].toInt(), sv[15], sv[8], d["R"].toInt(), sv[9], sv[10], sv[11], false,
                                               sv[16], d["track_off"].toBool(), d["map_off"].toBool(), need);
        consume_post_backward_wait(stream_of(dev));
        // the reference sums a [H*W,4,4] buffer over dim 0 (L/__init__.py:160-161); here it is [1,4,4], already reduced
        Tensor gview = view_of(g[8], 0, {4, 4}, at::kFloat);
        variable_list out(25);
        out[0] = std::move(g[3]); out[1] = std::move(g[0]); out[2] = std::move(g[5]); out[3] = std::move(g[1]);
        out[4] = std::move(g[2]); out[5] = std::move(g[6]); out[6] = std::move(g[7]); out[7] = std::move(g[4]);
        out[8] = std::move(gview);
        return out;
    }
};

// (color, radii, depth, depth_median, depth_var, opacity_map, gau_uncertainty, gau_related_pixels), num_rendered or -1,
// ticket or -1, capacity used, device status word
std::tuple<std::vector<Tensor>, long, long, long, Tensor>
light_apply(const Tensor& means3D, const Tensor& means2D, const Tensor& sh, const Tensor& colors_precomp, const Tensor& opacities,
            const Tensor& scales, const Tensor& rotations, const Tensor& cov3Ds_precomp, const Tensor& viewmatrix,
            const Tensor& gt_depth, const Tensor& bg, const Tensor& projmatrix, const Tensor& campos, const Tensor& perspec,
            double scale_modifier, double tanfovx, double tanfovy, long H, long W, long degree, bool prefiltered, bool track_off,
            bool map_off, long capacity, long mode) {
    Probe p_a(HP_APPLY);
    variable_list out = LightNode::apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, viewmatrix,
                                         gt_depth, bg, projmatrix, campos, perspec, scale_modifier, tanfovx, tanfovy, (int64_t)H,
                                         (int64_t)W, (int64_t)degree, prefiltered, track_off, map_off, (int64_t)capacity, (int64_t)mode);
    p_a.stop();
    Tensor status = std::move(g_report.status);
    g_report.status = Tensor();
    return {std::move(out), g_report.rendered, g_report.ticket, g_report.cap, std::move(status)};
}

struct FullNode : public torch::autograd::Function<FullNode> {
    static variable_list forward(AutogradContext* ctx, const Tensor& means3D, const Tensor& means2D, const Tensor& sh,
                                 const Tensor& colors_precomp, const Tensor& opacities, const Tensor& scales, const Tensor& rotations,
                                 const Tensor& cov3Ds_precomp, const Tensor& viewmatrix, const Tensor& gt_depth, const Tensor& bg,
                                 const Tensor& projmatrix, const Tensor& campos, const Tensor& perspec, double scale_modifier,
                                 double tanfovx, double tanfovy, int64_t H, int64_t W, int64_t degree, bool prefiltered,
                                 int64_t capacity, int64_t mode) {
        (void)means2D;
        FullFwd o = full_forward_core(bg, means3D, colors_precomp, opacities, scales, rotations, scale_modifier, cov3Ds_precomp,
                                      viewmatrix, gt_depth, projmatrix, tanfovx, tanfovy, H, W, sh, degree, campos, prefiltered,
                                      capacity, mode, /*want_related=*/false);
        g_report.rendered = o.rendered; g_report.related = o.related; g_report.ticket = o.ticket; g_report.cap = o.cap;
        g_report.status = o.status;
        // F/__init__.py:89-90
        ctx->save_for_backward({colors_precomp, means3D, scales, rotations, cov3Ds_precomp, viewmatrix, o.radii, sh, o.geom,
                                o.binning, o.img, gt_depth, bg, projmatrix, campos, perspec});
        auto& d = ctx->saved_data;
        d["scale_modifier"] = scale_modifier; d["tanfovx"] = tanfovx; d["tanfovy"] = tanfovy; d["degree"] = degree;
        d["R"] = (int64_t)(o.rendered >= 0 ? o.rendered : o.cap); d["H"] = H; d["W"] = W;
        d["options"] = (int64_t)dgr_thread_options_effective();
        ctx->set_materialize_grads(false);
        ctx->mark_non_differentiable({o.radii});
        return {o.color, o.radii, o.depth, o.unc};
    }

    static variable_list backward(AutogradContext* ctx, variable_list grad) {
        const auto sv = ctx->get_saved_variables();
        auto& d = ctx->saved_data;
        const Tensor& means3D = sv[1];
        const c10::Device dev = means3D.device();
        const long H = d["H"].toInt(), W = d["W"].toInt();
        const Tensor gC = grad[0].defined() ? grad[0] : zeros_like_image(3, H, W, dev);
        const Tensor gD = grad[2].defined() ? grad[2] : zeros_like_image(1, H, W, dev);
        // (no gradient image for the uncertainty output: NULL at the C ABI, which then runs the lean blend backward)
        const Tensor gU = grad[3].defined() ? grad[3] : Tensor();
        bool need = false;
        for (int i = 0; i < 8; i++) need = need || ctx->needs_input_grad(i);
        const UnderOptions under((int)d["options"].toInt());
        std::vector<Tensor> g = full_backward(sv[12], means3D, sv[6], sv[0], sv[2], sv[3], d["scale_modifier"].toDouble(), sv[4], sv[5],
                                              sv[11], sv[13], d["tanfovx"].toDouble(), d["tanfovy"].toDouble(), gC, gD, gU, sv[7],
                                              d["degree"].toInt(), sv[14], sv[8], d["R"].toInt(), sv[9], sv[10], 0, sv[15], need);
        consume_post_backward_wait(stream_of(dev));
        variable_list out(23);
        out[0] = std::move(g[3]); out[1] = std::move(g[0]); out[2] = std::move(g[5]); out[3] = std::move(g[1]);
        out[4] = std::move(g[2]); out[5] = std::move(g[6]); out[6] = std::move(g[7]); out[7] = std::move(g[4]);
        out[8] = std::move(g[8]);
        return out;
    }
};

// (color, radii, depth, uncertainty), num_rendered or -1, num_related or -1, ticket or -1, capacity used, device status word
std::tuple<std::vector<Tensor>, long, long, long, long, Tensor>
full_apply(const Tensor& means3D, const Tensor& means2D, const Tensor& sh, const Tensor& colors_precomp, const Tensor& opacities,
           const Tensor& scales, const Tensor& rotations, const Tensor& cov3Ds_precomp, const Tensor& viewmatrix,
           const Tensor& gt_depth, const Tensor& bg, const Tensor& projmatrix, const Tensor& campos, const Tensor& perspec,
           double scale_modifier, double tanfovx, double tanfovy, long H, long W, long degree, bool prefiltered, long capacity,
           long mode) {
    variable_list out = FullNode::apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, viewmatrix,
                                        gt_depth, bg, projmatrix, campos, perspec, scale_modifier, tanfovx, tanfovy, (int64_t)H,
                                        (int64_t)W, (int64_t)degree, prefiltered, (int64_t)capacity, (int64_t)mode);
    Tensor status = std::move(g_report.status);
    g_report.status = Tensor();
    return {std::move(out), g_report.rendered, g_report.related, g_report.ticket, g_report.cap, std::move(status)};
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
                                         bool need_gaussian_grads, bool need_means2D, const std::vector<long>& num_rendered) {
    const c10::Device dev = means3D_.device();
    c10::hip::HIPGuardMasqueradingAsCUDA guard(dev);
    const int P = (int)means3D_.size(0);
    const long V = viewmatrices_.size(0), H = dL_dout_color.size(2), W = dL_dout_color.size(3);
    if (V < 1 || V > DGR_MAX_BATCH_VIEWS) throw std::runtime_error("1 .. " + std::to_string(DGR_MAX_BATCH_VIEWS) + " views per batch");
    const Tensor means3D = f32c(means3D_, dev), bg = f32c(background, dev), colors = f32c(colors_, dev),
                 scales = f32c(scales_, dev), rotations = f32c(rotations_, dev), cov3D = f32c(cov3D_, dev),
                 views = f32c(viewmatrices_, dev), projs = f32c(projmatrices_, dev), campos = f32c(campos_, dev),
                 gts = f32c(gt_depths_, dev), sh = f32c(sh_, dev), alphas = f32c(alphas_, dev), perspec = f32c_diag4(perspec_, dev),
                 gC = f32c(dL_dout_color, dev), gD = f32c(dL_dout_depth, dev), gM = f32c(dL_dout_median, dev),
                 gV = f32c(dL_dout_var, dev);
    const int M = sh.numel() != 0 ? (int)sh.size(1) : 0;
    const auto f32 = at::TensorOptions().dtype(at::kFloat).device(dev);
    std::vector<Tensor> g(9);
    float* gp[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    Tensor d2;
    if (need_gaussian_grads) {
        grad_arena(dev, P, M, g.data());
        g[0].zero_();  // the arena's one-view means2D slot: a batch returns those gradients per view, beside the arena
        g[0] = Tensor();
        for (int i = 1; i < 8; i++) gp[i] = ptr<float>(g[i]);
        if (need_means2D) { d2 = at::empty({V, P, 3}, f32); g[0] = d2; }
    } else {
        map_off = true;  // nobody reads the per-Gaussian sums: the blend kernels form the three pose sums only
    }
    Tensor dview = at::empty({V, 4, 4}, f32);
    // (deterministic_grads: + 64 bytes per tile instance of the view with the most of them; the views' rows are equally long)
    long rmax = 0;
    for (long r : num_rendered) rmax = std::max(rmax, r);
    const size_t nscr = std::max<size_t>(up256(dgr_light_backward_scratch_bytes_r(P, (int)W, (int)H, (int)rmax)), 256);
    Tensor scratch = at::empty({V, (long long)nscr}, at::TensorOptions().dtype(at::kByte).device(dev));
    dgr_light_view_grad w[DGR_MAX_BATCH_VIEWS];
    for (long v = 0; v < V; v++) {
        w[v] = dgr_light_view_grad{row_bytes(geom, v), row_bytes(binning, v), row_bytes(img, v), row<float>(views, v),
                                   row<float>(projs, v), row<float>(campos, v), ptr<float>(perspec), row<float>(alphas, v),
                                   row<float>(gts, v), row<int>(radii, v), row<float>(gC, v), row<float>(gD, v), row<float>(gM, v),
                                   row<float>(gV, v), d2.defined() ? row<float>(d2, v) : nullptr, row<float>(dview, v),
                                   row_bytes(scratch, v), nscr, (size_t)v < num_rendered.size() ? (int)num_rendered[v] : 0};
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
    m.def("host_prof_dump", &host_prof_dump);
    m.def("light_apply", &light_apply);
    m.def("full_apply", &full_apply);
    m.def("set_post_backward_wait", &set_post_backward_wait);
    m.def("drop_post_backward_wait", &drop_post_backward_wait);
    m.def("mark_visible", &mark_visible);
    m.def("status_poll", &status_poll);
    // which kind of views the outputs are: 1 = raw TensorImpl windows, 0 = dispatcher views, -1 = not decided yet (no view made)
    m.def("raw_views", [] { return g_raw_views.load(); });
    // TEST-ONLY (tests/test_hip_binding_guard.py): not synchronised with forwards in flight on other threads
    m.def("set_raw_views", [](long v) { g_raw_views.store(v < 0 ? -1 : v ? 1 : 0); });
}
