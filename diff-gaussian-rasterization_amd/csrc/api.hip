// api.hip -- C-ABI entry points (include/dgr_hip.h) and host orchestration.
//
// Replaces CudaRasterizer::Rasterizer::{forward, backward, markVisible}
// (L/cuda_rasterizer/rasterizer_impl.cu:141-153, 197-350, 354-495).  Stage order per view:
//   forward : zero histogram -> preprocess (+ per-tile histogram) -> scan tiles -> emit keys ->
//             per-tile sort -> blend
//   backward: zero accumulator rows -> blend backward -> fused per-Gaussian backward -> pose reduce
// Nothing here touches the CPU oracle; a missing GPU or a failed launch is reported, never papered over.
#include <hip/hip_runtime.h>

#include <chrono>
#include <climits>
#include <cstdlib>
#include <mutex>
#include <utility>
#include <vector>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <mutex>
#include <string>
#include <thread>

#include "../../include/dgr_hip.h"
#include "dgr_common.h"
#include "kernels.h"

namespace dgr {
thread_local LaunchEvents* g_launch_events = nullptr;
}
namespace { extern std::atomic<int> g_blend_wgs_per_cu; }
namespace dgr {
size_t blend_pad_bytes(const void* kernel) {
    const int n = g_blend_wgs_per_cu.load(std::memory_order_relaxed);
    if (n < 3 || n > 7) return 0;
    static std::mutex mu;
    static std::vector<std::pair<const void*, size_t>> known;  // static LDS bytes of the blend kernels seen so far
    size_t static_lds = ~(size_t)0;
    {
        std::lock_guard<std::mutex> lk(mu);
        for (const auto& e : known)
            if (e.first == kernel) static_lds = e.second;
        if (static_lds == ~(size_t)0) {
            hipFuncAttributes attr{};
            if (hipFuncGetAttributes(&attr, kernel) != hipSuccess) return 0;
            static_lds = attr.sharedSizeBytes;
            known.emplace_back(kernel, static_lds);
        }
    }
    // the smallest LDS claim that keeps workgroup n + 1 off a CU: n claims then leave the rest of the 160 KB to whatever else
    // fits beside them (160 / n each, as through round 6, left nothing -- and every front-end kernel stages through LDS)
    const size_t per = (((size_t)(160 * 1024) / (size_t)(n + 1)) & ~(size_t)255) + 256;
    return per > static_lds ? per - static_lds : 0;
}
}

namespace {

thread_local std::string g_last_error = "";

int hip_fail(hipError_t e, const char* what) {
    g_last_error = std::string(what) + ": " + hipGetErrorString(e);
    return DGR_ERR_HIP;
}
#define HIP_TRY(expr)                                   \
    do {                                                \
        hipError_t _e = (expr);                         \
        if (_e != hipSuccess) return hip_fail(_e, #expr); \
    } while (0)

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---- early status (dgr_early_status_arm / _wait): num_rendered and the prefiltered flag are final after scan_blocks,
// a tenth of the way into the forward; a caller that needs them on the host (the reference's blocking copy of
// num_rendered) waits for a copy issued at that point instead of for the whole forward.
struct EarlyStatus {
    bool armed = false, pending = false;
    hipEvent_t ev = nullptr;   // an event belongs to the device that was current when it was created:
    int ev_device = -1;        // re-created when this thread moves to another device
    int* pinned = nullptr;
};
thread_local EarlyStatus g_early;

int early_status_post(const int* device_status, hipStream_t st) {
    if (!g_early.armed) return DGR_OK;
    g_early.armed = false;
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (!g_early.ev || g_early.ev_device != dev) {
        if (g_early.ev) HIP_TRY(hipEventDestroy(g_early.ev));
        g_early.ev = nullptr;
        HIP_TRY(hipEventCreateWithFlags(&g_early.ev, hipEventDisableTiming));
        g_early.ev_device = dev;
    }
    if (!g_early.pinned) HIP_TRY(hipHostMalloc((void**)&g_early.pinned, 4 * sizeof(int), hipHostMallocDefault));
    HIP_TRY(hipMemcpyAsync(g_early.pinned, device_status, 4 * sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipEventRecord(g_early.ev, st));
    g_early.pending = true;
    return DGR_OK;
}

// Waiting for a status copy that is tens of microseconds away: hipEventSynchronize parks the thread and pays a wake-up
// of the order of 100 us when the event has not fired yet (measured: a 640x480 tracking iteration went from 0.46 to
// 0.55 ms when the status moved 20 us later in the forward), so poll for a while first.
hipError_t wait_event_spinning(hipEvent_t ev) {
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t e = hipEventQuery(ev);
        if (e != hipErrorNotReady) return e;
        if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(400)) return hipEventSynchronize(ev);
    }
}

// ---- resident backward scratch (dgr_backward_scratch_clean_arm): the next backward of this thread finds its scratch all zero and
// leaves it all zero -- no clearing launch in front of the blend backward.
thread_local bool g_scratch_clean_armed = false;

// ---- asynchronous status read-back (dgr_status_post / _poll): the lazy mode of the bindings copies a forward's status
// word to pinned host memory behind an event and looks at it one or two calls later.  Slots are pooled per device.
// Two ways to fill a slot: dgr_status_post copies a device word behind an event (any status word, after the fact);
// dgr_status_arm hands the slot to the NEXT presized forward, whose forward blend (workgroup 0, first thing) writes the word
// straight into the slot's pinned memory (mapped into the device's address space) with a tag last -- no copy, no event, nothing
// to wait for on the stream.  One device word owned by the slot (zero between forwards) gathers the frame's longest tile list.
struct StatusSlot {
    hipEvent_t ev = nullptr;
    int* pinned = nullptr;       // host int[8]: {num_rendered, overflow, prefiltered violation, num_related | tag, longest list, -, -}
    int* pinned_dev = nullptr;   // the same memory as the device sees it
    uint32_t* ws = nullptr;      // device uint32[16], zero between forwards
    int device = -1;
    bool busy = false;
    bool mapped = false;         // this use of the slot: armed (written by the kernels) rather than posted (copied)
    uint32_t tag = 0;
    int W = 0, H = 0, P = 0;     // the forward that took the arm (key of the schedule hint below)
    hipStream_t stream = nullptr;  // ... and the stream its kernels were enqueued on (dgr_status_poll watches it while it waits)
    bool enqueued = false;         // the forward's blend kernel -- which delivers the word -- has been enqueued
    bool quarantined = false;      // a poll gave this slot up (timeout, stream error) while its forward may still be queued: the
                                   // blend kernel can still write words 0-5 and its tag here, so the slot is not handed out again
                                   // before that stream has drained (status_slot_acquire)
};
std::mutex g_status_mu;
std::vector<StatusSlot> g_status_slots;
uint32_t g_status_tag = 0;
thread_local long g_armed_slot = -1;

// ---- tile schedule policy (dgr_set_option("tile_schedule", v)): 1 = every forward runs tile_schedule_kernel (the blend kernels
// take their tiles classes of long lists first), 0 = never (static XCD band map), 2 (default) = by the frame: a forward whose
// status word came back through an armed slot also reports its longest tile list, and the NEXT forward of that shape
// (device, P, W, H) skips the schedule when the longest list was within 2x the mean + 32 -- on such a frame the schedule buys
// nothing (uniform synth-v1 scene: 1 % of the blend time) and costs a launch, a 1024-thread workgroup in front of the blend
// (11 us at 1080p) and some of the blend's L2 locality; a clustered frame (longest list 5x the mean) gets it back one
// forward later.  Forwards without a report (callback path, batched entry points, hipGraph capture, direct C-ABI callers that
// never arm) keep the schedule.  Results do not depend on it: only the order in which tiles are worked on.
std::atomic<int> g_tile_schedule{[] { const char* e = getenv("DGR_TILE_SCHEDULE"); return (e && e[0] >= '0' && e[0] <= '2') ? e[0] - '0' : 2; }()};
struct SchedHint { int device, W, H, P, on, longest; };
std::vector<SchedHint> g_sched_hints;  // (under g_status_mu)
bool want_schedule(int W, int H, int P) {
    const int mode = g_tile_schedule.load(std::memory_order_relaxed);
    if (mode != 2) return mode != 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return true;
    std::lock_guard<std::mutex> lk(g_status_mu);
    for (const auto& h : g_sched_hints)
        if (h.device == dev && h.W == W && h.H == H && h.P == P) return h.on != 0;
    return true;
}
// The same report also sizes the binning's row segments (segment_binning.hip: segment_shift): the longest tile list of this shape's
// last reported frame, or -1 without one.  On a clustered frame the capacity alone says "16 tiles per segment" (the AVERAGE
// segment fits bin_tiles' LDS) while every segment of the cluster overflows it and takes the dense path.
int hinted_longest_list(int W, int H, int P) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    std::lock_guard<std::mutex> lk(g_status_mu);
    for (const auto& h : g_sched_hints)
        if (h.device == dev && h.W == W && h.H == H && h.P == P) return h.longest;
    return -1;
}
void note_schedule_hint(const StatusSlot& sl, const int* word) {  // (g_status_mu held)
    const long tiles = (long)dgr::tiles_x(sl.W) * dgr::tiles_y(sl.H);
    if (tiles <= 0 || word[1] /* overflow: the lists were left empty */) return;
    const long longest = word[5];
    const int on = (longest >= 0x7fffffff || longest * tiles > 2L * word[0] + 32L * tiles) ? 1 : 0;
    const int lg = longest >= 0x7fffffff ? -1 : (int)longest;
    for (auto& h : g_sched_hints)
        if (h.device == sl.device && h.W == sl.W && h.H == sl.H && h.P == sl.P) { h.on = on; h.longest = lg; return; }
    if (g_sched_hints.size() >= 64) g_sched_hints.erase(g_sched_hints.begin());
    g_sched_hints.push_back(SchedHint{sl.device, sl.W, sl.H, sl.P, on, lg});
}

// The armed slot of this thread, taken by a presized forward.  If the call leaves before its binning kernel is enqueued
// (an error, P == 0) the word is completed from the host -- all zero -- so that a poll never waits for a write that will not come.
struct ArmedReport {
    long id = -1;
    dgr::StatusReport rep{nullptr, 0u, nullptr};
    bool handed_over = false;
    ArmedReport(int W, int H, int P, hipStream_t st = nullptr) {
        id = g_armed_slot;
        g_armed_slot = -1;
        if (id < 0) return;
        std::lock_guard<std::mutex> lk(g_status_mu);
        StatusSlot& sl = g_status_slots[(size_t)id];
        sl.W = W; sl.H = H; sl.P = P; sl.stream = st; sl.enqueued = false;
        rep.host = sl.pinned_dev; rep.tag = sl.tag; rep.ws = sl.ws;
    }
    ~ArmedReport() {
        if (id < 0) return;
        std::lock_guard<std::mutex> lk(g_status_mu);
        StatusSlot& sl = g_status_slots[(size_t)id];
        if (handed_over) { sl.enqueued = true; return; }
        volatile int* w = sl.pinned;
        w[0] = w[1] = w[2] = w[3] = 0; w[5] = 0x7fffffff;
        w[4] = (int)sl.tag;
    }
};

// ---- optional per-stage timing with HIP events on the launching stream (dgr_profile_* in dgr_hip.h).
// Disabled by default; when a stage is selected, two events bracket that stage's launch only.
struct StageProf {
    const char* name;
    bool on = false;
    unsigned seen = 0;  // launches of this stage since it was selected
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
};
StageProf g_prof[] = {{"zero_counters"}, {"preprocess_fwd"}, {"scan_blocks"}, {"bin_segments"}, {"bin_tiles"}, {"count_rank"},
                      {"scan_tiles"}, {"emit_instances"}, {"sort_tiles"}, {"tile_schedule"}, {"render_fwd"}, {"zero_scratch"}, {"render_bwd"},
                      {"preprocess_bwd"}};
enum { ST_ZERO_FWD, ST_PRE_FWD, ST_SCAN_BLOCKS, ST_BIN_SEGMENTS, ST_BIN_TILES, ST_COUNT_RANK, ST_SCAN, ST_EMIT, ST_SORT, ST_TILE_SCHED, ST_RENDER_FWD,
       ST_ZERO, ST_RENDER_BWD, ST_PRE_BWD, ST_COUNT };
std::mutex g_prof_mu;
std::atomic<int> g_profile_every{1};

// dgr_set_option("tight_cull", 1): alpha-aware tile rectangles (preprocess.hip); process-wide, default off
std::atomic<int> g_tight_cull{0};
// dgr_set_option("alpha_mode", v): how the blend kernels evaluate alpha and T / (1 - alpha) (csrc/render_common.h).
//   0 (default) = the reference's expression with the CPU restatement's bits (exp_p32 / div_ref, csrc/exact_math.h): alpha
//       image, n_contrib and median depth bit-identical to the restatement, gradients within 1e-5 of it;
//   1 (= "fast_alpha", 1) = log2(e)-scaled conic, one v_exp_f32, v_rcp_f32: every operation good to an ulp, but the light
//       backward's T_final = 1 - alpha and its divisions by (1 - alpha) amplify the last-bit differences to 6e-5 abs at config 3;
//   2 = as 0 with glibc's expf algorithm in the double pipe (exp_glibc; rounds 5-7's default, the oracle's exp mode 1), for A/B.
// Set it before the forward whose backward should use it (forward and backward of a view must use the same mode).
// Initial value from DGR_ALPHA_MODE (or DGR_FAST_ALPHA=1), for A/B runs.
std::atomic<int> g_alpha_mode{[] {
    const char* m = getenv("DGR_ALPHA_MODE");
    if (m && m[0] >= '0' && m[0] <= '2' && m[1] == 0) return m[0] - '0';
    const char* e = getenv("DGR_FAST_ALPHA");
    return (e && e[0] == '1') ? 1 : 0;
}()};
// dgr_set_option("deterministic_grads", 1): the light backward (one-view entry point, alpha_mode 0) forms its gradients without
// order-dependent float atomics (csrc/render_light.hip: DET): bit-identical run after run, at the price of an instance-major row
// buffer (64 bytes per tile instance: zero-filled, written and read once) and a smaller batch in the blend backward.  The backward
// then needs dgr_light_backward_scratch_bytes_r(P, W, H, R) bytes of scratch, R = the value passed as `R` (>= num_rendered).
std::atomic<int> g_det_grads{[] { const char* e = getenv("DGR_DETERMINISTIC_GRADS"); return (e && e[0] == '1') ? 1 : 0; }()};

// dgr_set_option("lane_lists", v): the lists the LIGHT blend kernels walk (csrc/render_light.hip).
//   1 = one list per half of a quadrant wave in the forward and the tracking backward, paired lists in the mapping backward (round 8);
//   0 = one list per quadrant wave everywhere (rounds 1-7);
//   2 (default) = decided per FRAME on the device by the binning kernel, from the frame's own run statistics (segment_binning.hip:
//       bin_tiles_kernel; big splats -> 0) and recorded in the frame's state, where forward and backward read it.
// Initial value from DGR_FWD_HALVES = 0 / 1 (the switch's name when it was per process; A/B runs).
std::atomic<int> g_lane_lists{[] { const char* e = getenv("DGR_FWD_HALVES"); return (e && (e[0] == '0' || e[0] == '1') && e[1] == 0) ? e[0] - '0' : 2; }()};

// ---- per-THREAD overrides of the three options that change what a call computes (dgr_set_thread_option, round 9).  The options
// above are process-wide defaults; a tracker thread and a mapper thread of one process -- or a test beside a training loop -- hold
// their own values here (-1 = inherit).  Every entry point reads its options ONCE, when it is called, and hands them to its
// launches as template choices / kernel arguments: launches already queued (on any stream) are not affected by a later change.
// A backward must run with its forward's alpha mode: the autograd bindings snapshot dgr_thread_options_effective() in the
// forward and swap it in around the backward (which the autograd engine may run on another thread).
thread_local int t_alpha_mode = -1, t_tight_cull = -1, t_det_grads = -1;
inline int opt_alpha_mode() { return t_alpha_mode >= 0 ? t_alpha_mode : g_alpha_mode.load(std::memory_order_relaxed); }
inline int opt_tight_cull() { return t_tight_cull >= 0 ? t_tight_cull : g_tight_cull.load(std::memory_order_relaxed); }
inline int opt_det_grads() { return t_det_grads >= 0 ? t_det_grads : g_det_grads.load(std::memory_order_relaxed); }
// dgr_set_option("lds_count", v): how the forward bins tile instances.
//   1 (default) = the two-level segment binning (csrc/segment_binning.hip) whenever the frame's segment tables fit LDS;
//   0 = returning global atomics on per-tile counters (csrc/binning.hip; inside preprocess_fwd when presized), which also
//       serves frames too large for the segment tables.  (2 is accepted as a synonym of 1.)
// Measured, round 5 (profiles/r5/): the segment binning is faster one view at a time at every size (config 3: 0.51 against
// 0.59 ms per view, config 4: 1.71 / 1.85, config 5: 4.54 / 4.86) and equal or faster with three views in flight (0.455 /
// 0.466, 1.57 / 1.57, 4.27 / 4.53), so nothing switches by job size or by the number of views in flight any more.
// Initial value from DGR_LDS_COUNT (for A/B runs).
std::atomic<int> g_lds_count{[] { const char* e = getenv("DGR_LDS_COUNT"); return (e && e[0] >= '0' && e[0] <= '2') ? e[0] - '0' : 1; }()};

// dgr_set_option("blend_wgs_per_cu", n): cap on the blend kernels' workgroups per CU (kernels.h: launch_blend); 0 = none.
// Initial value from DGR_BLEND_WGS_PER_CU (for A/B runs).
std::atomic<int> g_blend_wgs_per_cu{[] { const char* e = getenv("DGR_BLEND_WGS_PER_CU"); return (e && e[0] >= '3' && e[0] <= '7') ? e[0] - '0' : 0; }()};

// A kernel stage hands its two events to the stage's first kernel launch (dgr::launch, kernels.h): they then hold
// that kernel's start and end.  A stage without a kernel (the scratch memset) is bracketed with hipEventRecord.
struct ScopedStage {
    StageProf* p = nullptr;
    hipStream_t st;
    dgr::LaunchEvents le{};
    bool kernel_stage;
    ScopedStage(int id, hipStream_t s, bool is_kernel = true) : st(s), kernel_stage(is_kernel) {
        if (!g_prof[id].on) return;
        // dgr_set_option("profile_every", n): bracket every n-th launch only (the events ride in the dispatch packet
        // and cost a little overlap between streams; a sample keeps the timed region undisturbed)
        if (g_prof[id].seen++ % (unsigned)std::max(1, g_profile_every.load()) != 0) return;
        p = &g_prof[id];
        if (hipEventCreate(&le.start) != hipSuccess || hipEventCreate(&le.stop) != hipSuccess) { p = nullptr; return; }
        le.used = false;
        if (kernel_stage) dgr::g_launch_events = &le;
        else (void)hipEventRecord(le.start, st);
    }
    ~ScopedStage() {
        if (!p) return;
        if (kernel_stage) {
            dgr::g_launch_events = nullptr;
            if (!le.used) {  // nothing was launched (empty input)
                (void)hipEventDestroy(le.start);
                (void)hipEventDestroy(le.stop);
                return;
            }
        } else {
            (void)hipEventRecord(le.stop, st);
        }
        std::lock_guard<std::mutex> lk(g_prof_mu);
        p->ev.emplace_back(le.start, le.stop);
    }
};

struct FwdCommon {
    int P, D, M, W, H;
    const float *background, *means3D, *shs, *colors_precomp, *opacities, *scales, *rotations, *cov3D_precomp;
    float scale_modifier;
    const float *viewmatrix, *projmatrix, *cam_pos;
    float tan_fovx, tan_fovy;
    int prefiltered;
    float *out_color, *out_depth, *out_median_depth, *out_alpha;
    const float* gt_depth;
    float *out_depth_var, *gau_uncertainty;
    int *gau_related_pixels, *radii;
};

// P == 0: the reference launches nothing and returns zero-filled outputs (L/rasterize_points.cu:88).
int zero_outputs(const FwdCommon& c, hipStream_t st) {
    const size_t N = (size_t)c.W * c.H;
    HIP_TRY(hipMemsetAsync(c.out_color, 0, 3 * N * 4, st));
    HIP_TRY(hipMemsetAsync(c.out_depth, 0, N * 4, st));
    if (c.out_median_depth) HIP_TRY(hipMemsetAsync(c.out_median_depth, 0, N * 4, st));
    if (c.out_alpha) HIP_TRY(hipMemsetAsync(c.out_alpha, 0, N * 4, st));
    if (c.out_depth_var) HIP_TRY(hipMemsetAsync(c.out_depth_var, 0, N * 4, st));
    return DGR_OK;
}

// preprocess.  Callback path (bin == nullptr): afterwards geom.block_tiles holds the instance totals per 256-Gaussian
// block and scan_blocks turns them into offsets and num_rendered.  Presized path (the binning buffer exists already):
// the kernel also takes the tile-counter atomics and stores the ranks (count_rank.h), behind one small clear of the
// counters; scan_blocks and count_rank disappear.
// Presized path, binning mode: COUNT_LDS = the two-level segment binning after preprocess (segment_binning.hip; frames
// whose segment tables fit LDS), COUNT_FUSED = returning global atomics inside preprocess_fwd (binning.hip).
// Callback path (the binning buffer is sized after a host read of num_rendered): COUNT_LDS_CALLBACK = the same segment
// binning behind scan_blocks, COUNT_CALLBACK = the count_rank kernel on global tile counters (R = 0, or a frame too large).
enum { COUNT_CALLBACK = 0, COUNT_FUSED = 1, COUNT_LDS = 2, COUNT_LDS_CALLBACK = 3 };
int presized_count_mode(int W, int H, int capacity) {
    const int v = g_lds_count.load();
    (void)capacity;
    const bool lds = v != 0 && dgr::segment_binning_fits(W, H);
    return lds ? COUNT_LDS : COUNT_FUSED;
}
int forward_front(const FwdCommon& c, dgr::GeometryView geom, dgr::ImageView img, hipStream_t st,
                  const dgr::BinningView* bin = nullptr, int capacity = 0, char* image_base = nullptr,
                  int mode = COUNT_CALLBACK) {
    const int gx = dgr::tiles_x(c.W), gy = dgr::tiles_y(c.H), tiles = gx * gy;
    // No memsets: preprocess clears the per-Gaussian median statistics (and, on the callback path, the tile counters);
    // scan_blocks / scan_tiles initialise the status word.
    dgr::PreprocessFwdArgs a{};
    a.P = c.P; a.D = c.D; a.M = c.M; a.W = c.W; a.H = c.H; a.grid_x = gx; a.grid_y = gy;
    a.means3D = c.means3D; a.scales = c.scales; a.scale_modifier = c.scale_modifier; a.rotations = c.rotations;
    a.opacities = c.opacities; a.shs = c.shs; a.cov3D_precomp = c.cov3D_precomp; a.colors_precomp = c.colors_precomp;
    a.view = c.viewmatrix; a.proj = c.projmatrix; a.campos = c.cam_pos;
    a.tan_fovx = c.tan_fovx; a.tan_fovy = c.tan_fovy;
    a.focal_y = c.H / (2.0f * c.tan_fovy);  // rasterizer_impl.cu:228-229
    a.focal_x = c.W / (2.0f * c.tan_fovx);
    a.prefiltered = c.prefiltered;
    a.tight_cull = opt_tight_cull();
    a.sh_vec_ok = aligned16(c.shs);
    a.geom = geom; a.radii_out = c.radii;
    a.gau_uncertainty = c.gau_uncertainty; a.gau_related_pixels = c.gau_related_pixels;
    (void)tiles;
    if (bin && mode == COUNT_LDS) {
        // nothing to clear: the kernel leaves its per-block instance totals (and the `prefiltered` flag) in
        // geom.block_tiles, bin_segments / bin_tiles take it from there
        { ScopedStage t(ST_PRE_FWD, st); HIP_TRY(dgr::launch_preprocess_fwd(a, st)); }
        return DGR_OK;
    }
    if (bin) {
        // cursor + padded tile counters: everything between the start of the image buffer and the range table
        { ScopedStage t(ST_ZERO_FWD, st); HIP_TRY(dgr::launch_zero_fill(image_base, (size_t)((char*)img.ranges - image_base), st)); }
        a.fused_count = 1; a.tile_count = img.tile_count; a.cursor = img.cursor; a.ranks = bin->ranks; a.capacity = capacity;
        { ScopedStage t(ST_PRE_FWD, st); HIP_TRY(dgr::launch_preprocess_fwd(a, st)); }
        return DGR_OK;
    }
    a.zero_words = img.tile_count; a.n_zero_words = (int)(((char*)img.ranges - (char*)img.tile_count) / 4);
    { ScopedStage t(ST_PRE_FWD, st); HIP_TRY(dgr::launch_preprocess_fwd(a, st)); }
    // per-block instance totals -> exclusive prefix; status[0] = num_rendered
    { ScopedStage t(ST_SCAN_BLOCKS, st); HIP_TRY(dgr::launch_scan_blocks(c.P, geom, img, st)); }
    return DGR_OK;
}

// histogram + ranks, range table (status[0] = num_rendered, status[1] = overflow), key scatter, per-tile sort
// (`fused`: preprocess_fwd counted already; the status word is complete after scan_tiles, which is where a caller that
// armed the early status gets its copy)
int binning_stages(const FwdCommon& c, dgr::GeometryView geom, dgr::ImageView img, dgr::BinningView bin, int capacity,
                   hipStream_t st, int mode = COUNT_CALLBACK, char* binning_base = nullptr, ArmedReport* armed = nullptr) {
    const int gx = dgr::tiles_x(c.W), gy = dgr::tiles_y(c.H), tiles = gx * gy;
    // the tile schedule: always, unless this shape's last reported frame had even lists (want_schedule above)
    const bool sched_on = !(armed && armed->id >= 0) ? (g_tile_schedule.load(std::memory_order_relaxed) != 0) : want_schedule(c.W, c.H, c.P);
    const int lists = g_lane_lists.load(std::memory_order_relaxed);
    const int blend_flags = (sched_on ? dgr::BLEND_SCHEDULE : 0) | (lists == 0 ? dgr::BLEND_LISTS_QUADRANT : lists == 2 ? dgr::BLEND_LISTS_AUTO : 0);
    const dgr::StatusReport rep = armed ? armed->rep : dgr::StatusReport{nullptr, 0u, nullptr};
    if (mode == COUNT_LDS || mode == COUNT_LDS_CALLBACK) {
        const bool cb = mode == COUNT_LDS_CALLBACK;
        const dgr::SegmentTables tb = dgr::carve_segment_tables(binning_base + bin.bytes, c.W, c.H);
        const int longest = (armed && armed->id >= 0) ? hinted_longest_list(c.W, c.H, c.P) : -1;
        const int ss = dgr::segment_shift(c.W, c.H, capacity, longest);
        { ScopedStage t(ST_BIN_SEGMENTS, st); HIP_TRY(dgr::launch_bin_segments(c.P, geom, bin, tb, gx, gy, ss, capacity, cb, st)); }
        { ScopedStage t(ST_BIN_TILES, st); HIP_TRY(dgr::launch_bin_tiles(c.P, geom, img, bin, tb, gx, gy, ss, capacity, cb, blend_flags, rep, st)); }
        if (!cb) { const int rc = early_status_post(img.status, st); if (rc) return rc; }  // (bin_tiles writes the status word)
        if (sched_on) { ScopedStage t(ST_TILE_SCHED, st); HIP_TRY(dgr::launch_tile_schedule(img, tiles, st)); }
        return DGR_OK;
    }
    const bool fused = mode == COUNT_FUSED;
    if (!fused) { ScopedStage t(ST_COUNT_RANK, st); HIP_TRY(dgr::launch_count_rank(c.P, geom, img, bin, gx, capacity, st)); }
    { ScopedStage t(ST_SCAN, st); HIP_TRY(dgr::launch_scan_tiles(img, tiles, gx, capacity, fused, blend_flags, rep, st)); }
    if (fused) { const int rc = early_status_post(img.status, st); if (rc) return rc; }
    { ScopedStage t(ST_EMIT, st); HIP_TRY(dgr::launch_emit_instances(c.P, geom, img, bin, gx, st)); }
    { ScopedStage t(ST_SORT, st); HIP_TRY(dgr::launch_sort_tiles(img, bin, tiles, st)); }
    if (sched_on) { ScopedStage t(ST_TILE_SCHED, st); HIP_TRY(dgr::launch_tile_schedule(img, tiles, st)); }
    return DGR_OK;
}

int forward_back(const FwdCommon& c, dgr::GeometryView geom, dgr::ImageView img, dgr::BinningView bin, hipStream_t st,
                 ArmedReport* armed = nullptr) {
    const int gx = dgr::tiles_x(c.W), gy = dgr::tiles_y(c.H);
    dgr::RenderFwdLightArgs r{};
    r.W = c.W; r.H = c.H; r.grid_x = gx; r.grid_y = gy;
    r.sched = img.tile_sched; r.ranges = img.ranges; r.sched_flag = img.cursor + 3; r.point_list = bin.point_list; r.rec = geom.rec; r.bg = c.background; r.gt_depth = c.gt_depth;
    r.out_color = c.out_color; r.out_depth = c.out_depth; r.out_median = c.out_median_depth; r.out_alpha = c.out_alpha;
    r.out_depth_var = c.out_depth_var; r.n_contrib = img.n_contrib; r.gau_uncertainty = c.gau_uncertainty;
    r.gau_related_pixels = c.gau_related_pixels;
    r.rep = armed ? armed->rep : dgr::StatusReport{nullptr, 0u, nullptr};
    r.status = img.status;
    { ScopedStage t(ST_RENDER_FWD, st); HIP_TRY(dgr::launch_render_fwd_light(r, opt_alpha_mode(), st)); }
    if (armed) armed->handed_over = true;  // (workgroup 0 of the blend delivers the word)
    return DGR_OK;
}

// ---- full variant: same front end, different blend
int forward_back_full(const FwdCommon& c, float* out_uncertainty, dgr::GeometryView geom, dgr::ImageView img,
                      dgr::BinningView bin, hipStream_t st, ArmedReport* armed = nullptr) {
    const int gx = dgr::tiles_x(c.W), gy = dgr::tiles_y(c.H);
    dgr::RenderFwdFullArgs r{};
    r.W = c.W; r.H = c.H; r.grid_x = gx; r.grid_y = gy;
    r.sched = img.tile_sched; r.ranges = img.ranges; r.sched_flag = img.cursor + 3; r.point_list = bin.point_list; r.rec = geom.rec; r.bg = c.background;
    r.out_color = c.out_color; r.out_depth = c.out_depth; r.out_uncertainty = out_uncertainty;
    r.n_contrib = img.n_contrib; r.n_valid = img.n_valid; r.first_contrib = img.first_contrib; r.final_T = img.final_T;
    r.status = img.status;
    r.rep = armed ? armed->rep : dgr::StatusReport{nullptr, 0u, nullptr};
    { ScopedStage t(ST_RENDER_FWD, st); HIP_TRY(dgr::launch_render_fwd_full(r, opt_alpha_mode(), st)); }
    if (armed) armed->handed_over = true;
    return DGR_OK;
}

int check_common(const FwdCommon& c) {
    if (c.P < 0 || c.W <= 0 || c.H <= 0) { g_last_error = "bad sizes"; return DGR_ERR_BAD_ARGUMENT; }
    if ((unsigned)c.P > DGR_ID_MASK) { g_last_error = "more than 2^28 Gaussians"; return DGR_ERR_BAD_ARGUMENT; }
    if (c.P > 0 && !c.shs && !c.colors_precomp) { g_last_error = "need SHs or precomputed colours"; return DGR_ERR_BAD_ARGUMENT; }
    if (c.P > 0 && !c.cov3D_precomp && (!c.scales || !c.rotations)) { g_last_error = "need scale/rotation or cov3D"; return DGR_ERR_BAD_ARGUMENT; }
    // (2^30 pixels: the blend kernels index pixels, and the three colour planes, with 32-bit words)
    if (dgr::tiles_x(c.W) > 65535 || dgr::tiles_y(c.H) > 65535 || (long long)c.W * c.H > (1ll << 30)) { g_last_error = "image too large"; return DGR_ERR_BAD_ARGUMENT; }
    return DGR_OK;
}

// ---- batched views (dgr_light_forward_batch / dgr_light_backward_batch) ----
// The per-Gaussian kernels of a batch run ONCE for all views on the caller's stream; the per-view stages in between
// (binning + blend, blend backward) are independent and mix kernels that leave the chip idle (count, scan, emit, sort) with
// kernels bound by VALU issue (blend), so view v runs them on stream v mod K -- the caller's stream and K - 1 helper
// streams of this thread -- forked and joined with events: what DESIGN.md s7 measured for "views in flight", inside one call
// and capturable into one hipGraph.  dgr_set_option("batch_streams", K), K = 1..8: at most K streams; the views are dealt
// out in rounds, and a batch uses the fewest streams that keep the number of rounds minimal, since the join waits for the
// longest stream.  Default 2, measured (profiles/batch_streams.sh, config 3, ms per view for K = 1 / 2 / 3 / 4 / 8):
// 3 views 0.459 / 0.434 / 0.436 / 0.438 / 0.444, 4 views 0.450 / 0.415 / 0.419 / 0.446 / 0.437, 8 views 0.425 / 0.377 /
// 0.380 / 0.391 / 0.407 -- one view's binning under another view's blend is the whole gain; more blend kernels at once
// only take each other's L2 and wave slots.
std::atomic<int> g_batch_streams{2};
// dgr_set_option("batch_order", o): how the per-view stages of a batch are spread over the streams.
//   0 (default) = round robin: view v's whole chain on stream v mod K ("batch_streams");
//   1 = pipeline: the views' BINNING stages (count, scan, emit, sort -- kernels that leave most of the chip idle) one after the
//       other on a helper stream, the views' BLEND stages one after the other on the caller's stream, view v's blend waiting
//       for view v's binning with an event (the backward likewise: the next view's scratch cleared under the current blend).
//       On paper binning v + 1 always runs under blend v and two VALU-bound blend kernels never share the chip; measured
//       (profiles/batch_order.sh) it LOSES to the round robin at every size -- config 3, ms per view, pipeline / round robin:
//       2 views 0.509 / 0.461, 4 views 0.468 / 0.417, 8 views 0.431 / 0.383; config 2: 0.193 / 0.156, 0.172 / 0.130,
//       0.156 / 0.116 -- slower even than one stream (0.450 at 4 views): one cross-stream event wait per view each way costs more
//       than the overlap it arranges (the same finding as round 1's high-priority companion stream).  Kept as the
//       measured alternative.
std::atomic<int> g_batch_order{0};
constexpr int DGR_BATCH_MAX_STREAMS = 8;
inline int batch_stream_count(int n_views) {
    const int kmax = std::max(1, std::min(g_batch_streams.load(), DGR_BATCH_MAX_STREAMS));
    const int rounds = (n_views + kmax - 1) / kmax;
    return (n_views + rounds - 1) / rounds;
}
struct BatchStreams {
    int device = -1;
    hipStream_t helper[DGR_BATCH_MAX_STREAMS - 1] = {};
    hipEvent_t fork = nullptr;
    hipEvent_t join[DGR_BATCH_MAX_STREAMS - 1] = {};
    hipEvent_t stage[DGR_MAX_BATCH_VIEWS] = {};  // pipelined order: view v's binning (forward) / cleared scratch (backward) is ready
};
// One set per device and thread: streams and events belong to the device that was current when they were created, and a
// single-process loop that alternates between GPUs (dgr_amd._capi.on_device) must neither re-create them on every call
// nor lose the old ones.
constexpr int DGR_BATCH_MAX_DEVICES = 32;
thread_local BatchStreams g_batch_pool[DGR_BATCH_MAX_DEVICES];
thread_local BatchStreams* g_batch_p = nullptr;
int batch_streams_ready() {
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= DGR_BATCH_MAX_DEVICES) { g_last_error = "device index above DGR_BATCH_MAX_DEVICES"; return DGR_ERR_BAD_ARGUMENT; }
    g_batch_p = &g_batch_pool[dev];
    if (g_batch_p->device == dev) return DGR_OK;
    for (int i = 0; i < DGR_BATCH_MAX_STREAMS - 1; i++) {
        HIP_TRY(hipStreamCreateWithFlags(&g_batch_p->helper[i], hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&g_batch_p->join[i], hipEventDisableTiming));
    }
    HIP_TRY(hipEventCreateWithFlags(&g_batch_p->fork, hipEventDisableTiming));
    for (int v = 0; v < DGR_MAX_BATCH_VIEWS; v++) HIP_TRY(hipEventCreateWithFlags(&g_batch_p->stage[v], hipEventDisableTiming));
    g_batch_p->device = dev;
    return DGR_OK;
}
// stream of view v among K; fork: the helpers wait for what the caller's stream has enqueued so far
inline hipStream_t batch_stream(hipStream_t main, int v, int K) { return (v % K == 0) ? main : g_batch_p->helper[v % K - 1]; }
int batch_fork(hipStream_t main, int K) {
    if (K <= 1) return DGR_OK;
    HIP_TRY(hipEventRecord(g_batch_p->fork, main));
    for (int i = 0; i < K - 1; i++) HIP_TRY(hipStreamWaitEvent(g_batch_p->helper[i], g_batch_p->fork, 0));
    return DGR_OK;
}
int batch_join(hipStream_t main, int K) {
    for (int i = 0; i < K - 1; i++) {
        HIP_TRY(hipEventRecord(g_batch_p->join[i], g_batch_p->helper[i]));
        HIP_TRY(hipStreamWaitEvent(main, g_batch_p->join[i], 0));
    }
    return DGR_OK;
}
// After batch_fork the helper streams may hold kernels that touch the caller's buffers: whatever way the call leaves --
// an error in the middle of the view loop included -- the caller's stream must wait for them (and a stream capture must
// see the forked streams rejoined).  done() performs the join once and reports its status.
struct BatchJoinGuard {
    hipStream_t main;
    int K;
    bool armed;
    BatchJoinGuard(hipStream_t m, int k) : main(m), K(k), armed(true) {}
    int done() { armed = false; return batch_join(main, K); }
    ~BatchJoinGuard() { if (armed) batch_join(main, K); }
};

// ---- state export (tests / profiling) ----
enum ExportKind { EX_MEANS2D, EX_CONIC_OPACITY, EX_RGB, EX_CLAMPED, EX_TILES_TOUCHED, EX_KEYS };

__global__ void export_geom_kernel(int kind, int P, dgr::GeometryView g, void* dst) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const bool vis = g.radii[i] > 0;
    switch (kind) {
        case EX_MEANS2D: {
            const float4 q = g.rec[DGR_REC_STRIDE * (size_t)i];
            ((float2*)dst)[i] = vis ? make_float2(q.x, q.y) : make_float2(0, 0);
        } break;
        case EX_CONIC_OPACITY: {
            const float4 q0 = g.rec[DGR_REC_STRIDE * (size_t)i], q1 = g.rec[DGR_REC_STRIDE * (size_t)i + 1];
            ((float4*)dst)[i] = vis ? make_float4(q1.x, q1.y, q1.z, q0.w) : make_float4(0, 0, 0, 0);
        } break;
        case EX_RGB: {
            const float4 q = g.rec[DGR_REC_STRIDE * (size_t)i + 2];
            float* d = (float*)dst + 3 * (size_t)i;
            d[0] = vis ? q.x : 0; d[1] = vis ? q.y : 0; d[2] = vis ? q.z : 0;
        } break;
        case EX_CLAMPED: {
            const uint8_t c = vis ? g.clamped[i] : 0;
            uint8_t* d = (uint8_t*)dst + 3 * (size_t)i;
            d[0] = c & 1; d[1] = (c >> 1) & 1; d[2] = (c >> 2) & 1;
        } break;
        case EX_TILES_TOUCHED: {
            const ushort4 r = g.rect[i];
            ((uint32_t*)dst)[i] = (uint32_t)(r.z - r.x) * (uint32_t)(r.w - r.y);
        } break;
    }
}
// the reference's sorted 64-bit keys: tile id << 32 | depth bits (rasterizer_impl.cu:97-100)
__global__ void export_keys_kernel(dgr::ImageView img, dgr::BinningView bin, dgr::GeometryView g, uint64_t* dst) {
    const int tile = blockIdx.x;
    const uint2 rg = img.ranges[tile];
    for (uint32_t i = rg.x + threadIdx.x; i < rg.y; i += blockDim.x)
        dst[i] = ((uint64_t)tile << 32) | __float_as_uint(g.depths[bin.point_list[i] & DGR_ID_MASK]);
}
// the sorted Gaussian ids (the mask: rounds 3-8 kept contribution tags in the top 4 bits; nothing writes them any more)
__global__ void export_point_list_kernel(const uint32_t* src, uint32_t* dst, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = src[i] & DGR_ID_MASK;
}
// ... and the contribution tags (tests): the blend forward's tag bytes (bit 2 w + h: half h of quadrant wave w; `half` = 1: as they
// are), or folded to 4 bits, bit w = quadrant wave w
__global__ void export_tag_bytes_kernel(const uint8_t* src, uint8_t* dst, int n, int half) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint32_t t = src[i];
    if (!half) {
        t = (t | (t >> 1)) & 0x55u;
        t = (t & 1u) | ((t >> 1) & 2u) | ((t >> 2) & 4u) | ((t >> 3) & 8u);
    }
    dst[i] = (uint8_t)t;
}

}  // namespace

extern "C" {

const char* dgr_last_error(void) { return g_last_error.c_str(); }
const char* dgr_version(void) { return "dgr_hip 0.1 gfx950"; }

size_t dgr_geometry_bytes(int P) { return dgr::carve_geometry(nullptr, P).bytes; }
size_t dgr_image_bytes(int width, int height) { return dgr::carve_image(nullptr, width, height).bytes; }
size_t dgr_binning_bytes(int cap, int width, int height) {
    // the sorted list, key scratch, ranks / pair columns and pair keys of `cap` instances, then the forward-only tables of
    // the segment binning
    return dgr::carve_binning(nullptr, (size_t)(cap > 0 ? cap : 0)).bytes + dgr::carve_segment_tables(nullptr, width, height).bytes;
}
size_t dgr_light_backward_scratch_bytes(int P, int, int) { return dgr::carve_backward_scratch(nullptr, P).bytes; }
namespace {
// deterministic gradients: behind the standard scratch, {per-block instance sums u32[blocks] | per-block pose partials
// double[blocks][12] | instance-major rows float[R][16]}
struct DetScratch {
    uint32_t* blk;
    double* pose;
    float* rows;
    size_t bytes;
};
DetScratch carve_det_scratch(char* base, int P, int R) {
    const size_t nb = ((size_t)(P > 0 ? P : 0) + 255) / 256;
    DetScratch d;
    size_t o = dgr::carve_backward_scratch(nullptr, P).bytes;
    d.blk = (uint32_t*)(base + o);  o = dgr::align_up(o + 4 * nb, 256);
    d.pose = (double*)(base + o);   o = dgr::align_up(o + 8 * 12 * nb, 256);
    d.rows = (float*)(base + o);    o = dgr::align_up(o + sizeof(float) * DGR_ACC_STRIDE * (size_t)(R > 0 ? R : 0), 256);
    d.bytes = o;
    return d;
}
}  // namespace
size_t dgr_light_backward_scratch_bytes_r(int P, int W, int H, int R) {
    if (!opt_det_grads()) return dgr_light_backward_scratch_bytes(P, W, H);
    return carve_det_scratch(nullptr, P, R).bytes;
}

int dgr_mark_visible(void* stream, int P, const float* means3D, const float* viewmatrix, const float*, uint8_t* present) {
    HIP_TRY(dgr::launch_mark_visible(P, means3D, viewmatrix, present, (hipStream_t)stream));
    return DGR_OK;
}

int dgr_light_forward_presized(void* stream, char* geometry_buffer, char* binning_buffer, int binning_capacity,
                               char* image_buffer, int* status, int P, int D, int M, const float* background,
                               int width, int height, const float* means3D, const float* shs,
                               const float* colors_precomp, const float* opacities, const float* scales,
                               float scale_modifier, const float* rotations, const float* cov3D_precomp,
                               const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                               float tan_fovx, float tan_fovy, int prefiltered, float* out_color, float* out_depth,
                               float* out_median_depth, float* out_alpha, const float* gt_depth,
                               float* out_depth_var, float* gau_uncertainty, int* gau_related_pixels, int* radii) {
    hipStream_t st = (hipStream_t)stream;
    FwdCommon c{P, D, M, width, height, background, means3D, shs, colors_precomp, opacities, scales, rotations,
                cov3D_precomp, scale_modifier, viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy, prefiltered,
                out_color, out_depth, out_median_depth, out_alpha, gt_depth, out_depth_var, gau_uncertainty,
                gau_related_pixels, radii};
    ArmedReport armed(width, height, P, st);  // (dgr_status_arm: completed from the host on every path that enqueues no binning kernel)
    int rc = check_common(c);
    if (rc) return rc;
    if (P == 0) {
        if (status) HIP_TRY(hipMemsetAsync(status, 0, 16, st));
        return zero_outputs(c, st);
    }
    // (the binning buffer also holds the segment binning's tables behind its per-instance arrays: dgr_binning_bytes() is
    //  non-zero for a capacity of 0, and a NULL buffer is never valid for P > 0)
    if (!geometry_buffer || !image_buffer || !binning_buffer || binning_capacity < 0) {
        g_last_error = "presized forward: geometry, binning and image buffers are required (sizes: dgr_*_bytes)";
        return DGR_ERR_BAD_ARGUMENT;
    }
    dgr::GeometryView geom = dgr::carve_geometry(geometry_buffer, P);
    dgr::ImageView img = dgr::carve_image(image_buffer, width, height);
    if (status) img.status = status;  // the kernels write the caller's status word directly
    dgr::BinningView bin = dgr::carve_binning(binning_buffer, (size_t)binning_capacity);
    const int mode = presized_count_mode(width, height, binning_capacity);
    if ((rc = forward_front(c, geom, img, st, &bin, binning_capacity, image_buffer, mode))) return rc;
    if ((rc = binning_stages(c, geom, img, bin, binning_capacity, st, mode, binning_buffer, &armed))) return rc;
    if ((rc = forward_back(c, geom, img, bin, st, &armed))) return rc;
    return DGR_OK;
}

int dgr_light_forward(void* stream, dgr_alloc_fn geometryBuffer, dgr_alloc_fn binningBuffer, dgr_alloc_fn imageBuffer,
                      void* alloc_user, int P, int D, int M, const float* background, int width, int height,
                      const float* means3D, const float* shs, const float* colors_precomp, const float* opacities,
                      const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
                      const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx,
                      float tan_fovy, int prefiltered, float* out_color, float* out_depth, float* out_median_depth,
                      float* out_alpha, const float* gt_depth, float* out_depth_var, float* gau_uncertainty,
                      int* gau_related_pixels, int* radii, int debug) {
    hipStream_t st = (hipStream_t)stream;
    FwdCommon c{P, D, M, width, height, background, means3D, shs, colors_precomp, opacities, scales, rotations,
                cov3D_precomp, scale_modifier, viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy, prefiltered,
                out_color, out_depth, out_median_depth, out_alpha, gt_depth, out_depth_var, gau_uncertainty,
                gau_related_pixels, radii};
    int rc = check_common(c);
    if (rc) return rc;
    // The callback entry points block the host to size the binning buffer, as the reference does (rasterizer_impl.cu:287): on a
    // capturing stream that synchronisation would fail AND invalidate the capture -- refuse before anything touches the stream.
    if (dgr_stream_is_capturing(stream)) {
        g_last_error = "the resize-callback forward blocks the host (it sizes the binning buffer): it cannot be captured into a graph -- use the presized entry point";
        return DGR_ERR_BAD_ARGUMENT;
    }
    if (P == 0) return zero_outputs(c, st);
    char* gptr = geometryBuffer(dgr_geometry_bytes(P), alloc_user);
    char* iptr = imageBuffer(dgr_image_bytes(width, height), alloc_user);
    if (!gptr || !iptr) { g_last_error = "allocation callback returned NULL"; return DGR_ERR_ALLOC; }
    dgr::GeometryView geom = dgr::carve_geometry(gptr, P);
    dgr::ImageView img = dgr::carve_image(iptr, width, height);
    if ((rc = forward_front(c, geom, img, st))) return rc;
    // the one blocking read the reference also has (rasterizer_impl.cu:287): num_rendered sizes the binning buffer
    int status[4] = {0, 0, 0, 0};
    HIP_TRY(hipMemcpyAsync(status, img.status, sizeof(status), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (status[2]) { g_last_error = "Point is filtered although prefiltered is set. This shouldn't happen!"; return DGR_ERR_PREFILTERED; }
    const int R = status[0];
    char* bptr = nullptr;
    if (R > 0) {
        bptr = binningBuffer(dgr_binning_bytes(R, width, height), alloc_user);
        if (!bptr) { g_last_error = "allocation callback returned NULL"; return DGR_ERR_ALLOC; }
    } else {
        binningBuffer(0, alloc_user);
    }
    dgr::BinningView bin = dgr::carve_binning(bptr, (size_t)R);
    // (also with R == 0: it writes the (empty) range table)
    const int mode = (R > 0 && presized_count_mode(width, height, R) == COUNT_LDS) ? COUNT_LDS_CALLBACK : COUNT_CALLBACK;
    if ((rc = binning_stages(c, geom, img, bin, R, st, mode, bptr))) return rc;
    if ((rc = forward_back(c, geom, img, bin, st))) return rc;
    if (debug) HIP_TRY(hipStreamSynchronize(st));  // CHECK_CUDA(..., debug): L/cuda_rasterizer/auxiliary.h:166-173
    return R;
}

int dgr_light_backward(void* stream, int P, int D, int M, int R, const float* background, int width, int height,
                       const float* means3D, const float* shs, const float* colors_precomp, const float* alphas,
                       const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
                       const float* viewmatrix, const float* projmatrix, const float* campos, float tan_fovx,
                       float tan_fovy, const int* radii, char* geom_buffer, char* binning_buffer, char* image_buffer,
                       const float* dL_dpix, const float* dL_dpix_depth, const float* dL_dpix_median_depth,
                       const float* dL_dpix_depth_var, float* dL_dmean2D, float* dL_dconic, float* dL_dopacity,
                       float* dL_dcolor, float* dL_ddepth, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh,
                       float* dL_dscale, float* dL_drot, int debug, float* dgndcs_dviewmatrix,
                       const float* perspec_matrix, float* dL_dview, float* dg_camd_dviewmatrix,
                       const float* gt_depth, int track_off, int map_off, char* scratch, size_t scratch_bytes) {
    (void)dgndcs_dviewmatrix; (void)dg_camd_dviewmatrix; (void)colors_precomp;
    hipStream_t st = (hipStream_t)stream;
    const bool scratch_clean = g_scratch_clean_armed;
    g_scratch_clean_armed = false;
    if (P < 0 || width <= 0 || height <= 0 || (long long)width * height > (1ll << 30)) { g_last_error = "bad sizes"; return DGR_ERR_BAD_ARGUMENT; }
    if (P == 0) {  // L/rasterize_points.cu:188: nothing runs, gradients stay zero
        HIP_TRY(hipMemsetAsync(dL_dview, 0, 16 * 4, st));
        return DGR_OK;
    }
    const bool det = opt_det_grads() != 0 && !(track_off && map_off);
    if (det && opt_alpha_mode() != 0) { g_last_error = "deterministic_grads needs alpha_mode 0"; return DGR_ERR_BAD_ARGUMENT; }
    if (det && R <= 0) { g_last_error = "deterministic_grads: the backward needs R >= num_rendered (it sizes the instance-major row buffer)"; return DGR_ERR_BAD_ARGUMENT; }
    if (scratch_bytes < dgr_light_backward_scratch_bytes_r(P, width, height, R) || !scratch) {
        g_last_error = det ? "backward scratch too small (deterministic_grads: dgr_light_backward_scratch_bytes_r)" : "backward scratch too small";
        return DGR_ERR_BAD_ARGUMENT;
    }
    // (the 3D covariance is not kept by the forward: the backward re-forms it from scale and rotation -- the SAME tensors the
    //  forward saw, or the bits differ -- unless the caller precomputed it)
    if (!cov3D_precomp && (!scales || !rotations)) { g_last_error = "backward: need scale/rotation or cov3D"; return DGR_ERR_BAD_ARGUMENT; }
    if (!geom_buffer || !binning_buffer || !image_buffer) { g_last_error = "backward: the forward's three state buffers are required"; return DGR_ERR_BAD_ARGUMENT; }
    dgr::GeometryView geom = dgr::carve_geometry(geom_buffer, P);
    dgr::ImageView img = dgr::carve_image(image_buffer, width, height);
    dgr::BackwardScratch sc = dgr::carve_backward_scratch(scratch, P);
    const int gx = dgr::tiles_x(width), gy = dgr::tiles_y(height);
    if (!scratch_clean) { ScopedStage t(ST_ZERO, st); HIP_TRY(dgr::launch_zero_fill(sc.acc, sc.zero_bytes, st)); }

    dgr::RenderBwdLightArgs r{};
    r.W = width; r.H = height; r.grid_x = gx; r.grid_y = gy;
    r.sched = img.tile_sched; r.ranges = img.ranges; r.sched_flag = img.cursor + 3; r.point_list = (const uint32_t*)binning_buffer; r.rec = geom.rec; r.bg = background;
    r.gt_depth = gt_depth; r.alphas = alphas; r.n_contrib = img.n_contrib; r.dL_dpix = dL_dpix;
    r.dL_dpix_depth = dL_dpix_depth; r.dL_dpix_median = dL_dpix_median_depth; r.dL_dpix_var = dL_dpix_depth_var;
    r.means3D = means3D; r.view = viewmatrix; r.acc = sc.acc; r.track_off = track_off; r.map_off = map_off;
    DetScratch ds{nullptr, nullptr, nullptr, 0};
    if (det) {
        // (the Gaussians' first-instance offsets go into the geometry state's goff array, which only the global-counter binning
        //  of the FORWARD uses: free by now)
        ds = carve_det_scratch(scratch, P, R);
        { ScopedStage t(ST_ZERO, st); HIP_TRY(dgr::launch_zero_fill(ds.rows, sizeof(float) * DGR_ACC_STRIDE * (size_t)R, st)); }
        HIP_TRY(dgr::launch_det_offsets(P, geom.rect, ds.blk, geom.goff, st));
        r.det_rows = ds.rows; r.det_rect = geom.rect; r.det_goff = geom.goff; r.det_R = (uint32_t)R;
    }
    { ScopedStage t(ST_RENDER_BWD, st); HIP_TRY(dgr::launch_render_bwd_light(r, opt_alpha_mode(), st)); }
    if (det) HIP_TRY(dgr::launch_det_gather(P, geom.rect, geom.goff, ds.rows, (uint32_t)R, sc.acc, st));

    dgr::PreprocessBwdArgs b{};
    b.det_pose = det ? ds.pose : nullptr;
    b.P = P; b.D = D; b.M = M; b.W = width; b.H = height; b.means3D = means3D; b.radii = radii ? radii : geom.radii; b.shs = shs;
    b.scales = scales;
    b.rotations = rotations; b.scale_modifier = scale_modifier; b.cov3D_precomp = cov3D_precomp; b.view = viewmatrix;
    b.proj = projmatrix; b.campos = campos; b.perspec = perspec_matrix; b.tan_fovx = tan_fovx; b.tan_fovy = tan_fovy;
    b.focal_y = height / (2.0f * tan_fovy);
    b.focal_x = width / (2.0f * tan_fovx);
    b.sh_vec_ok = aligned16(shs) && aligned16(dL_dsh);
    b.track_off = track_off; b.map_off = map_off; b.geom = geom; b.acc = sc.acc; b.clear_scratch = scratch_clean ? 1 : 0;
    b.dL_dmean2D = dL_dmean2D; b.dL_dconic = dL_dconic; b.dL_dopacity = dL_dopacity; b.dL_dcolor = dL_dcolor;
    b.dL_ddepth = dL_ddepth; b.dL_dmean3D = dL_dmean3D; b.dL_dcov3D = dL_dcov3D; b.dL_dsh = dL_dsh;
    b.dL_dscale = dL_dscale; b.dL_drot = dL_drot; b.pose_part = sc.pose_part; b.ticket = sc.ticket; b.dL_dview = dL_dview;
    { ScopedStage t(ST_PRE_BWD, st); HIP_TRY(dgr::launch_preprocess_bwd(b, st)); }
    if (debug && !dgr_stream_is_capturing(stream)) HIP_TRY(hipStreamSynchronize(st));  // (CHECK_CUDA(..., debug); a capturing stream cannot be waited for -- and the attempt would invalidate the capture)
    return DGR_OK;
}

// ------------------------------------------------------------------------------------------------ full variant
int dgr_full_forward_presized(void* stream, char* geometry_buffer, char* binning_buffer, int binning_capacity,
                              char* image_buffer, int* status, int P, int D, int M, const float* background, int width,
                              int height, const float* means3D, const float* shs, const float* colors_precomp,
                              const float* opacities, const float* scales, float scale_modifier, const float* rotations,
                              const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                              const float* cam_pos, float tan_fovx, float tan_fovy, int prefiltered, float* out_color,
                              float* out_depth, const float* gt_depth, float* out_uncertainty, int* radii) {
    hipStream_t st = (hipStream_t)stream;
    FwdCommon c{P, D, M, width, height, background, means3D, shs, colors_precomp, opacities, scales, rotations,
                cov3D_precomp, scale_modifier, viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy, prefiltered,
                out_color, out_depth, nullptr, out_uncertainty, gt_depth, nullptr, nullptr, nullptr, radii};
    ArmedReport armed(width, height, P, st);  // (dgr_status_arm: completed from the host on every path that enqueues no binning kernel)
    int rc = check_common(c);
    if (rc) return rc;
    if (P == 0) {
        if (status) HIP_TRY(hipMemsetAsync(status, 0, 16, st));
        return zero_outputs(c, st);
    }
    // (the binning buffer also holds the segment binning's tables behind its per-instance arrays: dgr_binning_bytes() is
    //  non-zero for a capacity of 0, and a NULL buffer is never valid for P > 0)
    if (!geometry_buffer || !image_buffer || !binning_buffer || binning_capacity < 0) {
        g_last_error = "presized forward: geometry, binning and image buffers are required (sizes: dgr_*_bytes)";
        return DGR_ERR_BAD_ARGUMENT;
    }
    dgr::GeometryView geom = dgr::carve_geometry(geometry_buffer, P);
    dgr::ImageView img = dgr::carve_image(image_buffer, width, height);
    if (status) img.status = status;  // the kernels write the caller's status word directly
    dgr::BinningView bin = dgr::carve_binning(binning_buffer, (size_t)binning_capacity);
    const int mode = presized_count_mode(width, height, binning_capacity);
    if ((rc = forward_front(c, geom, img, st, &bin, binning_capacity, image_buffer, mode))) return rc;
    if ((rc = binning_stages(c, geom, img, bin, binning_capacity, st, mode, binning_buffer, &armed))) return rc;
    if ((rc = forward_back_full(c, out_uncertainty, geom, img, bin, st, &armed))) return rc;
    return DGR_OK;
}

int dgr_full_forward(void* stream, dgr_alloc_fn geometryBuffer, dgr_alloc_fn binningBuffer, dgr_alloc_fn imageBuffer,
                     void* alloc_user, int P, int D, int M, const float* background, int width, int height,
                     const float* means3D, const float* shs, const float* colors_precomp, const float* opacities,
                     const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
                     const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx,
                     float tan_fovy, int prefiltered, float* out_color, float* out_depth, const float* gt_depth,
                     float* out_uncertainty, int* radii, int* num_related_primitives) {
    hipStream_t st = (hipStream_t)stream;
    FwdCommon c{P, D, M, width, height, background, means3D, shs, colors_precomp, opacities, scales, rotations,
                cov3D_precomp, scale_modifier, viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy, prefiltered,
                out_color, out_depth, nullptr, out_uncertainty, gt_depth, nullptr, nullptr, nullptr, radii};
    if (num_related_primitives) *num_related_primitives = 0;
    int rc = check_common(c);
    if (rc) return rc;
    // The callback entry points block the host to size the binning buffer, as the reference does (rasterizer_impl.cu:287): on a
    // capturing stream that synchronisation would fail AND invalidate the capture -- refuse before anything touches the stream.
    if (dgr_stream_is_capturing(stream)) {
        g_last_error = "the resize-callback forward blocks the host (it sizes the binning buffer): it cannot be captured into a graph -- use the presized entry point";
        return DGR_ERR_BAD_ARGUMENT;
    }
    if (P == 0) return zero_outputs(c, st);
    char* gptr = geometryBuffer(dgr_geometry_bytes(P), alloc_user);
    char* iptr = imageBuffer(dgr_image_bytes(width, height), alloc_user);
    if (!gptr || !iptr) { g_last_error = "allocation callback returned NULL"; return DGR_ERR_ALLOC; }
    dgr::GeometryView geom = dgr::carve_geometry(gptr, P);
    dgr::ImageView img = dgr::carve_image(iptr, width, height);
    if ((rc = forward_front(c, geom, img, st))) return rc;
    int status[4] = {0, 0, 0, 0};
    HIP_TRY(hipMemcpyAsync(status, img.status, sizeof(status), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));  // first blocking read of the reference (F/cuda_rasterizer/rasterizer_impl.cu:435)
    if (status[2]) { g_last_error = "Point is filtered although prefiltered is set. This shouldn't happen!"; return DGR_ERR_PREFILTERED; }
    const int R = status[0];
    char* bptr = nullptr;
    if (R > 0) {
        bptr = binningBuffer(dgr_binning_bytes(R, width, height), alloc_user);
        if (!bptr) { g_last_error = "allocation callback returned NULL"; return DGR_ERR_ALLOC; }
    } else {
        binningBuffer(0, alloc_user);
    }
    dgr::BinningView bin = dgr::carve_binning(bptr, (size_t)R);
    // (also with R == 0: it writes the (empty) range table)
    const int mode = (R > 0 && presized_count_mode(width, height, R) == COUNT_LDS) ? COUNT_LDS_CALLBACK : COUNT_CALLBACK;
    if ((rc = binning_stages(c, geom, img, bin, R, st, mode, bptr))) return rc;
    if ((rc = forward_back_full(c, out_uncertainty, geom, img, bin, st))) return rc;
    if (num_related_primitives) {  // second blocking read of the reference (:498)
        HIP_TRY(hipMemcpyAsync(status, img.status, sizeof(status), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        *num_related_primitives = status[3];
    }
    return R;
}

int dgr_full_backward(void* stream, int P, int D, int M, int R, const float* background, int width, int height,
                      const float* means3D, const float* shs, const float* colors_precomp, const float* scales,
                      float scale_modifier, const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                      const float* projmatrix, const float* campos, float tan_fovx, float tan_fovy, const int* radii,
                      char* geom_buffer, char* binning_buffer, char* image_buffer, const float* dL_dpix,
                      const float* dL_depths, float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor,
                      float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot,
                      float* dpixel_dgc, int* gau_id_list, int* pix_id_list, float* dgc_dCam_position, float* dpixel_dndcs,
                      const float* perspec_matrix, float* dgndcs_dviewmatrix, float* dpixel_dinvcovs,
                      float* dgc_invcovs_dT, float* dL_dview, float* dL_dgau_depth, float* ddepth_dndcs,
                      float* ddepth_dinvcovs, const float* gt_depth, const float* dL_duncertainties, char* scratch,
                      size_t scratch_bytes) {
    (void)colors_precomp; (void)dpixel_dgc; (void)gau_id_list; (void)pix_id_list; (void)dgc_dCam_position;
    (void)dpixel_dndcs; (void)dgndcs_dviewmatrix; (void)dpixel_dinvcovs; (void)dgc_invcovs_dT; (void)ddepth_dndcs;
    (void)ddepth_dinvcovs;
    const bool det = opt_det_grads() != 0;  // (round 9: the scheme of the light variant, csrc/render_light.hip: DET)
    if (det && opt_alpha_mode() != 0) { g_last_error = "deterministic_grads needs alpha_mode 0"; return DGR_ERR_BAD_ARGUMENT; }
    if (det && R <= 0) { g_last_error = "deterministic_grads: the backward needs R >= num_rendered (it sizes the instance-major row buffer)"; return DGR_ERR_BAD_ARGUMENT; }
    hipStream_t st = (hipStream_t)stream;
    const bool scratch_clean = g_scratch_clean_armed;
    g_scratch_clean_armed = false;
    if (P < 0 || width <= 0 || height <= 0 || (long long)width * height > (1ll << 30)) { g_last_error = "bad sizes"; return DGR_ERR_BAD_ARGUMENT; }
    if (P == 0) {
        HIP_TRY(hipMemsetAsync(dL_dview, 0, 16 * 4, st));
        return DGR_OK;
    }
    if (scratch_bytes < dgr_light_backward_scratch_bytes_r(P, width, height, R) || !scratch) {
        g_last_error = det ? "backward scratch too small (deterministic_grads: dgr_light_backward_scratch_bytes_r)" : "backward scratch too small";
        return DGR_ERR_BAD_ARGUMENT;
    }
    // (the 3D covariance is not kept by the forward: the backward re-forms it from scale and rotation -- the SAME tensors the
    //  forward saw, or the bits differ -- unless the caller precomputed it)
    if (!cov3D_precomp && (!scales || !rotations)) { g_last_error = "backward: need scale/rotation or cov3D"; return DGR_ERR_BAD_ARGUMENT; }
    if (!geom_buffer || !binning_buffer || !image_buffer) { g_last_error = "backward: the forward's three state buffers are required"; return DGR_ERR_BAD_ARGUMENT; }
    dgr::GeometryView geom = dgr::carve_geometry(geom_buffer, P);
    dgr::ImageView img = dgr::carve_image(image_buffer, width, height);
    dgr::BackwardScratch sc = dgr::carve_backward_scratch(scratch, P);
    const int gx = dgr::tiles_x(width), gy = dgr::tiles_y(height);
    if (!scratch_clean) { ScopedStage t(ST_ZERO, st); HIP_TRY(dgr::launch_zero_fill(sc.acc, sc.zero_bytes, st)); }
    DetScratch ds{nullptr, nullptr, nullptr, 0};
    if (det) {
        ds = carve_det_scratch(scratch, P, R);
        { ScopedStage t(ST_ZERO, st); HIP_TRY(dgr::launch_zero_fill(ds.rows, sizeof(float) * DGR_ACC_STRIDE * (size_t)R, st)); }
        HIP_TRY(dgr::launch_det_offsets(P, geom.rect, ds.blk, geom.goff, st));
    }

    dgr::RenderBwdFullArgs r{};
    r.W = width; r.H = height; r.grid_x = gx; r.grid_y = gy;
    r.sched = img.tile_sched; r.ranges = img.ranges; r.sched_flag = img.cursor + 3; r.point_list = (const uint32_t*)binning_buffer; r.rec = geom.rec; r.bg = background;
    r.gt_depth = gt_depth; r.final_T = img.final_T; r.n_contrib = img.n_contrib; r.first_contrib = img.first_contrib;
    r.dL_dpix = dL_dpix; r.dL_depths = dL_depths; r.dL_duncertainties = dL_duncertainties; r.acc = sc.acc;
    if (det) { r.det_rows = ds.rows; r.det_rect = geom.rect; r.det_goff = geom.goff; r.det_R = (uint32_t)R; }
    { ScopedStage t(ST_RENDER_BWD, st); HIP_TRY(dgr::launch_render_bwd_full(r, opt_alpha_mode(), st, det)); }
    if (det) HIP_TRY(dgr::launch_det_gather(P, geom.rect, geom.goff, ds.rows, (uint32_t)R, sc.acc, st));

    dgr::PreprocessBwdArgs b{};
    b.det_pose = det ? ds.pose : nullptr;
    b.P = P; b.D = D; b.M = M; b.means3D = means3D; b.radii = radii ? radii : geom.radii; b.shs = shs; b.scales = scales;
    b.rotations = rotations; b.scale_modifier = scale_modifier; b.cov3D_precomp = cov3D_precomp; b.view = viewmatrix;
    b.proj = projmatrix; b.campos = campos; b.perspec = perspec_matrix; b.tan_fovx = tan_fovx; b.tan_fovy = tan_fovy;
    b.focal_y = height / (2.0f * tan_fovy);
    b.focal_x = width / (2.0f * tan_fovx);
    b.sh_vec_ok = aligned16(shs) && aligned16(dL_dsh);
    b.track_off = 0; b.map_off = 0; b.full_variant = 1; b.geom = geom; b.acc = sc.acc; b.clear_scratch = scratch_clean ? 1 : 0;
    b.dL_dmean2D = dL_dmean2D; b.dL_dconic = dL_dconic; b.dL_dopacity = dL_dopacity; b.dL_dcolor = dL_dcolor;
    b.dL_ddepth = dL_dgau_depth; b.dL_dmean3D = dL_dmean3D; b.dL_dcov3D = dL_dcov3D; b.dL_dsh = dL_dsh;
    b.dL_dscale = dL_dscale; b.dL_drot = dL_drot; b.pose_part = sc.pose_part; b.ticket = sc.ticket; b.dL_dview = dL_dview;
    { ScopedStage t(ST_PRE_BWD, st); HIP_TRY(dgr::launch_preprocess_bwd(b, st)); }
    return DGR_OK;
}

int dgr_light_forward_batch(void* stream, int n_views, const dgr_light_view* views, int P, int D, int M,
                            const float* background, int width, int height, const float* means3D, const float* shs,
                            const float* colors_precomp, const float* opacities, const float* scales, float scale_modifier,
                            const float* rotations, const float* cov3D_precomp, float tan_fovx, float tan_fovy, int prefiltered) {
    hipStream_t st = (hipStream_t)stream;
    if (n_views < 1 || n_views > DGR_MAX_BATCH_VIEWS || !views) { g_last_error = "1 .. DGR_MAX_BATCH_VIEWS views per batch"; return DGR_ERR_BAD_ARGUMENT; }
    FwdCommon cv[DGR_MAX_BATCH_VIEWS];
    for (int v = 0; v < n_views; v++) {
        const dgr_light_view& w = views[v];
        cv[v] = FwdCommon{P, D, M, width, height, background, means3D, shs, colors_precomp, opacities, scales, rotations,
                          cov3D_precomp, scale_modifier, w.viewmatrix, w.projmatrix, w.cam_pos, tan_fovx, tan_fovy, prefiltered,
                          w.out_color, w.out_depth, w.out_median_depth, w.out_alpha, w.gt_depth, w.out_depth_var,
                          w.gau_uncertainty, w.gau_related_pixels, w.radii};
        if (!w.viewmatrix || !w.projmatrix || !w.cam_pos || !w.out_color || !w.out_depth) { g_last_error = "view without camera or outputs"; return DGR_ERR_BAD_ARGUMENT; }
        if (P > 0 && (!w.geometry_buffer || !w.image_buffer || !w.binning_buffer || w.binning_capacity < 0)) {
            g_last_error = "view without state buffers";
            return DGR_ERR_BAD_ARGUMENT;
        }
    }
    int rc = check_common(cv[0]);
    if (rc) return rc;
    if (P == 0) {
        for (int v = 0; v < n_views; v++) {
            if (views[v].status) HIP_TRY(hipMemsetAsync(views[v].status, 0, 16, st));
            if ((rc = zero_outputs(cv[v], st))) return rc;
        }
        return DGR_OK;
    }
    if ((rc = batch_streams_ready())) return rc;
    dgr::GeometryView geom[DGR_MAX_BATCH_VIEWS];
    dgr::ImageView img[DGR_MAX_BATCH_VIEWS];
    dgr::BinningView bin[DGR_MAX_BATCH_VIEWS];
    for (int v = 0; v < n_views; v++) {
        geom[v] = dgr::carve_geometry(views[v].geometry_buffer, P);
        img[v] = dgr::carve_image(views[v].image_buffer, width, height);
        if (views[v].status) img[v].status = views[v].status;
        bin[v] = dgr::carve_binning(views[v].binning_buffer, (size_t)views[v].binning_capacity);
    }
    const int gx = dgr::tiles_x(width), gy = dgr::tiles_y(height);
    // One preprocess launch for all views needs the segment binning behind it (its epilogue leaves per-block instance
    // totals); frames whose segment tables do not fit LDS, or "lds_count" = 0, take the one-view front end per view.
    const bool shared_front = g_lds_count.load() != 0 && dgr::segment_binning_fits(width, height);
    if (shared_front) {
        dgr::PreprocessFwdBatchArgs b{};
        dgr::PreprocessFwdArgs& a = b.base;
        a.P = P; a.D = D; a.M = M; a.W = width; a.H = height; a.grid_x = gx; a.grid_y = gy;
        a.means3D = means3D; a.scales = scales; a.scale_modifier = scale_modifier; a.rotations = rotations;
        a.opacities = opacities; a.shs = shs; a.cov3D_precomp = cov3D_precomp; a.colors_precomp = colors_precomp;
        a.tan_fovx = tan_fovx; a.tan_fovy = tan_fovy;
        a.focal_y = height / (2.0f * tan_fovy);  // rasterizer_impl.cu:228-229
        a.focal_x = width / (2.0f * tan_fovx);
        a.prefiltered = prefiltered;
        a.tight_cull = opt_tight_cull();
        a.sh_vec_ok = aligned16(shs);
        b.V = n_views;
        for (int v = 0; v < n_views; v++) {
            b.v[v].view = views[v].viewmatrix; b.v[v].proj = views[v].projmatrix; b.v[v].campos = views[v].cam_pos;
            b.v[v].geom = geom[v]; b.v[v].radii_out = views[v].radii; b.v[v].gau_uncertainty = views[v].gau_uncertainty;
            b.v[v].gau_related_pixels = views[v].gau_related_pixels;
        }
        { ScopedStage t(ST_PRE_FWD, st); HIP_TRY(dgr::launch_preprocess_fwd_batch(b, st)); }
    }
    const bool pipeline = g_batch_order.load() == 1 && n_views > 1 && g_batch_streams.load() > 1;
    const int K = pipeline ? 2 : batch_stream_count(n_views);
    if ((rc = batch_fork(st, K))) return rc;
    BatchJoinGuard joined(st, K);  // (an early return below still rejoins the helper streams)
    for (int v = 0; v < n_views; v++) {
        hipStream_t sv = pipeline ? g_batch_p->helper[0] : batch_stream(st, v, K);
        const int cap = views[v].binning_capacity;
        int mode = COUNT_LDS;
        if (!shared_front) {
            mode = presized_count_mode(width, height, cap);
            if ((rc = forward_front(cv[v], geom[v], img[v], sv, &bin[v], cap, views[v].image_buffer, mode))) return rc;
        }
        if ((rc = binning_stages(cv[v], geom[v], img[v], bin[v], cap, sv, mode, views[v].binning_buffer))) return rc;
        if (pipeline) {  // the blend of view v on the caller's stream, behind its binning on the helper stream
            HIP_TRY(hipEventRecord(g_batch_p->stage[v], sv));
            HIP_TRY(hipStreamWaitEvent(st, g_batch_p->stage[v], 0));
            sv = st;
        }
        if ((rc = forward_back(cv[v], geom[v], img[v], bin[v], sv))) return rc;
    }
    return joined.done();
}

int dgr_light_backward_batch(void* stream, int n_views, const dgr_light_view_grad* views, int P, int D, int M,
                             const float* background, int width, int height, const float* means3D, const float* shs,
                             const float* colors_precomp, const float* scales, float scale_modifier, const float* rotations,
                             const float* cov3D_precomp, float tan_fovx, float tan_fovy, float* dL_dopacity, float* dL_dcolor,
                             float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot,
                             int track_off, int map_off) {
    (void)colors_precomp;
    hipStream_t st = (hipStream_t)stream;
    const bool det = opt_det_grads() != 0 && !(track_off && map_off);  // (round 9: per view the scheme of the one-view backward)
    if (det && opt_alpha_mode() != 0) { g_last_error = "deterministic_grads needs alpha_mode 0"; return DGR_ERR_BAD_ARGUMENT; }
    if (n_views < 1 || n_views > DGR_MAX_BATCH_VIEWS || !views) { g_last_error = "1 .. DGR_MAX_BATCH_VIEWS views per batch"; return DGR_ERR_BAD_ARGUMENT; }
    if (P < 0 || width <= 0 || height <= 0 || (long long)width * height > (1ll << 30)) { g_last_error = "bad sizes"; return DGR_ERR_BAD_ARGUMENT; }
    for (int v = 0; v < n_views; v++)
        if (!views[v].dL_dview) { g_last_error = "view without dL_dview"; return DGR_ERR_BAD_ARGUMENT; }
    if (P > 0 && !cov3D_precomp && (!scales || !rotations)) { g_last_error = "backward: need scale/rotation or cov3D"; return DGR_ERR_BAD_ARGUMENT; }
    if (P == 0) {  // L/rasterize_points.cu:188: nothing runs, gradients stay zero
        for (int v = 0; v < n_views; v++) HIP_TRY(hipMemsetAsync(views[v].dL_dview, 0, 16 * 4, st));
        return DGR_OK;
    }
    for (int v = 0; v < n_views; v++) {
        const dgr_light_view_grad& w = views[v];
        if (det && w.num_rendered <= 0) { g_last_error = "deterministic_grads: every view needs num_rendered (it sizes the view's row buffer)"; return DGR_ERR_BAD_ARGUMENT; }
        const size_t need = dgr_light_backward_scratch_bytes_r(P, width, height, w.num_rendered);
        if (!w.scratch || w.scratch_bytes < need) { g_last_error = "backward scratch too small"; return DGR_ERR_BAD_ARGUMENT; }
        if (!w.geometry_buffer || !w.image_buffer || !w.viewmatrix || !w.projmatrix || !w.cam_pos || !w.perspec_matrix || !w.alphas ||
            !w.dL_dpix || !w.dL_dpix_depth || !w.dL_dpix_median_depth || !w.dL_dpix_depth_var) {
            g_last_error = "view with a missing state buffer, camera or gradient image";
            return DGR_ERR_BAD_ARGUMENT;
        }
    }
    int rc;
    if ((rc = batch_streams_ready())) return rc;
    const int gx = dgr::tiles_x(width), gy = dgr::tiles_y(height);
    const bool pipeline = g_batch_order.load() == 1 && n_views > 1 && g_batch_streams.load() > 1;
    const int K = pipeline ? 2 : batch_stream_count(n_views);
    dgr::PreprocessBwdBatchArgs bb{};
    if ((rc = batch_fork(st, K))) return rc;
    BatchJoinGuard joined(st, K);  // (an early return below still rejoins the helper streams)
    for (int v = 0; v < n_views; v++) {
        const dgr_light_view_grad& w = views[v];
        hipStream_t sv = pipeline ? g_batch_p->helper[0] : batch_stream(st, v, K);
        dgr::GeometryView geom = dgr::carve_geometry(w.geometry_buffer, P);
        dgr::ImageView img = dgr::carve_image(w.image_buffer, width, height);
        dgr::BackwardScratch sc = dgr::carve_backward_scratch(w.scratch, P);
        { ScopedStage t(ST_ZERO, sv); HIP_TRY(dgr::launch_zero_fill(sc.acc, sc.zero_bytes, sv)); }
        if (pipeline) {  // the blend backward of view v on the caller's stream, behind its cleared scratch
            HIP_TRY(hipEventRecord(g_batch_p->stage[v], sv));
            HIP_TRY(hipStreamWaitEvent(st, g_batch_p->stage[v], 0));
            sv = st;
        }
        dgr::RenderBwdLightArgs r{};
        r.W = width; r.H = height; r.grid_x = gx; r.grid_y = gy;
        r.sched = img.tile_sched; r.ranges = img.ranges; r.sched_flag = img.cursor + 3; r.point_list = (const uint32_t*)w.binning_buffer; r.rec = geom.rec; r.bg = background;
        r.gt_depth = w.gt_depth; r.alphas = w.alphas; r.n_contrib = img.n_contrib; r.dL_dpix = w.dL_dpix;
        r.dL_dpix_depth = w.dL_dpix_depth; r.dL_dpix_median = w.dL_dpix_median_depth; r.dL_dpix_var = w.dL_dpix_depth_var;
        r.means3D = means3D; r.view = w.viewmatrix; r.acc = sc.acc; r.track_off = track_off; r.map_off = map_off;
        DetScratch ds{nullptr, nullptr, nullptr, 0};
        if (det) {
            ds = carve_det_scratch(w.scratch, P, w.num_rendered);
            { ScopedStage t(ST_ZERO, sv); HIP_TRY(dgr::launch_zero_fill(ds.rows, sizeof(float) * DGR_ACC_STRIDE * (size_t)w.num_rendered, sv)); }
            HIP_TRY(dgr::launch_det_offsets(P, geom.rect, ds.blk, geom.goff, sv));
            r.det_rows = ds.rows; r.det_rect = geom.rect; r.det_goff = geom.goff; r.det_R = (uint32_t)w.num_rendered;
        }
        { ScopedStage t(ST_RENDER_BWD, sv); HIP_TRY(dgr::launch_render_bwd_light(r, opt_alpha_mode(), sv)); }
        if (det) HIP_TRY(dgr::launch_det_gather(P, geom.rect, geom.goff, ds.rows, (uint32_t)w.num_rendered, sc.acc, sv));
        dgr::BwdViewPart& q = bb.v[v];
        q.det_pose = det ? ds.pose : nullptr;
        q.view = w.viewmatrix; q.proj = w.projmatrix; q.campos = w.cam_pos; q.perspec = w.perspec_matrix;
        q.radii = w.radii ? w.radii : geom.radii; q.geom = geom; q.acc = sc.acc; q.dL_dmean2D = w.dL_dmean2D;
        q.pose_part = sc.pose_part; q.ticket = sc.ticket; q.dL_dview = w.dL_dview;
    }
    if ((rc = joined.done())) return rc;
    dgr::PreprocessBwdArgs& b = bb.base;
    b.P = P; b.D = D; b.M = M; b.W = width; b.H = height; b.means3D = means3D; b.shs = shs; b.scales = scales;
    b.rotations = rotations; b.scale_modifier = scale_modifier; b.cov3D_precomp = cov3D_precomp;
    b.tan_fovx = tan_fovx; b.tan_fovy = tan_fovy;
    b.focal_y = height / (2.0f * tan_fovy);
    b.focal_x = width / (2.0f * tan_fovx);
    b.sh_vec_ok = aligned16(shs) && aligned16(dL_dsh);
    b.track_off = track_off; b.map_off = map_off;
    b.dL_dopacity = dL_dopacity; b.dL_dcolor = dL_dcolor; b.dL_dmean3D = dL_dmean3D; b.dL_dcov3D = dL_dcov3D; b.dL_dsh = dL_dsh;
    b.dL_dscale = dL_dscale; b.dL_drot = dL_drot;
    bb.V = n_views;
    { ScopedStage t(ST_PRE_BWD, st); HIP_TRY(dgr::launch_preprocess_bwd_batch(bb, st)); }
    return DGR_OK;
}

int dgr_cov3d_forward(void* stream, int P, const float* scales, const float* rotations, float scale_modifier, float* cov3D) {
    if (P < 0 || (P > 0 && (!scales || !rotations || !cov3D))) { g_last_error = "dgr_cov3d_forward: bad argument"; return DGR_ERR_BAD_ARGUMENT; }
    HIP_TRY(dgr::launch_cov3d_forward(P, scales, rotations, scale_modifier, cov3D, (hipStream_t)stream));
    return DGR_OK;
}
int dgr_cov3d_backward(void* stream, int P, const float* scales, const float* rotations, float scale_modifier,
                       const float* dL_dcov3D, float* dL_dscales, float* dL_drotations) {
    if (P < 0 || (P > 0 && (!scales || !rotations || !dL_dcov3D || !dL_dscales || !dL_drotations))) {
        g_last_error = "dgr_cov3d_backward: bad argument";
        return DGR_ERR_BAD_ARGUMENT;
    }
    HIP_TRY(dgr::launch_cov3d_backward(P, scales, rotations, scale_modifier, dL_dcov3D, dL_dscales, dL_drotations, (hipStream_t)stream));
    return DGR_OK;
}

int dgr_debug_bin_tiles_trace(unsigned long long* device_words) {
    dgr::g_bin_tiles_trace = device_words;
    return DGR_OK;
}
int dgr_debug_wave_reduce(void* stream, const float* in, float* out16, float* out12, float* out4, int* comp16, int* comp12,
                          int* comp4) {
    HIP_TRY(dgr::launch_wave_reduce_test(in, out16, out12, out4, comp16, comp12, comp4, (hipStream_t)stream));
    return DGR_OK;
}
int dgr_debug_half_reduce(void* stream, const float* in, float* r0, float* r1, float* h3, int* slot0, int* slot1, int* comp3) {
    HIP_TRY(dgr::launch_half_reduce_test(in, r0, r1, h3, slot0, slot1, comp3, (hipStream_t)stream));
    return DGR_OK;
}
int dgr_debug_half_reduce16(void* stream, const float* in, float* r0, float* r1, int* slot0, int* slot1) {
    HIP_TRY(dgr::launch_half_reduce16_test(in, r0, r1, slot0, slot1, (hipStream_t)stream));
    return DGR_OK;
}
int dgr_debug_lane_lists(void* stream, const unsigned char* codes, unsigned* paired, unsigned* halves) {
    HIP_TRY(dgr::launch_lane_lists_test(codes, paired, halves, (hipStream_t)stream));
    return DGR_OK;
}
int dgr_debug_exact_math(void* stream, int n, const float* x, const float* a, const float* b, float* out_exp, float* out_div) {
    if (n < 0 || (n > 0 && (!x || !a || !b || !out_exp || !out_div))) {
        g_last_error = "dgr_debug_exact_math: bad argument";
        return DGR_ERR_BAD_ARGUMENT;
    }
    HIP_TRY(dgr::launch_exact_math_test(n, x, a, b, out_exp, out_div, opt_alpha_mode(), (hipStream_t)stream));
    return DGR_OK;
}

int dgr_sparse_adam(void* stream, long rows, int k, float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                    const int* visible, float lr, float beta1, float beta2, float eps, int step) {
    if (rows < 0 || k <= 0 || step < 1 || (rows > 0 && (!param || !grad || !exp_avg || !exp_avg_sq))) {
        g_last_error = "dgr_sparse_adam: bad argument";
        return DGR_ERR_BAD_ARGUMENT;
    }
    HIP_TRY(dgr::launch_sparse_adam((size_t)rows, k, param, grad, exp_avg, exp_avg_sq, visible, lr, beta1, beta2, eps, step,
                                    nullptr, (hipStream_t)stream));
    return DGR_OK;
}
int dgr_sparse_adam_capturable(void* stream, long rows, int k, float* param, const float* grad, float* exp_avg,
                               float* exp_avg_sq, const int* visible, float lr, float beta1, float beta2, float eps,
                               const int* step_device) {
    if (rows < 0 || k <= 0 || !step_device || (rows > 0 && (!param || !grad || !exp_avg || !exp_avg_sq))) {
        g_last_error = "dgr_sparse_adam_capturable: bad argument";
        return DGR_ERR_BAD_ARGUMENT;
    }
    HIP_TRY(dgr::launch_sparse_adam((size_t)rows, k, param, grad, exp_avg, exp_avg_sq, visible, lr, beta1, beta2, eps, 1,
                                    step_device, (hipStream_t)stream));
    return DGR_OK;
}

int dgr_densification_stats(void* stream, long rows, const float* dmeans2D, const int* radii, float* grad_accum, float* denom,
                            float* max_radii2D) {
    if (rows < 0 || rows > 0x7fffffffL || (rows > 0 && (!radii || (grad_accum && !dmeans2D)))) {
        g_last_error = "dgr_densification_stats: bad argument";
        return DGR_ERR_BAD_ARGUMENT;
    }
    HIP_TRY(dgr::launch_densification_stats((int)rows, dmeans2D, radii, grad_accum, denom, max_radii2D, (hipStream_t)stream));
    return DGR_OK;
}

int dgr_pose_forward(void* stream, const float* quat, const float* trans, const float* perspec_matrix, float* viewmatrix,
                     float* projmatrix, float* campos) {
    if (!quat || !trans || !perspec_matrix || !viewmatrix || !projmatrix || !campos) {
        g_last_error = "dgr_pose_forward: NULL argument";
        return DGR_ERR_BAD_ARGUMENT;
    }
    HIP_TRY(dgr::launch_pose_forward(quat, trans, perspec_matrix, viewmatrix, projmatrix, campos, (hipStream_t)stream));
    return DGR_OK;
}
int dgr_pose_backward(void* stream, const float* quat, const float* dL_dviewmatrix, float* dL_dquat, float* dL_dtrans) {
    if (!quat || !dL_dviewmatrix || !dL_dquat || !dL_dtrans) {
        g_last_error = "dgr_pose_backward: NULL argument";
        return DGR_ERR_BAD_ARGUMENT;
    }
    HIP_TRY(dgr::launch_pose_backward(quat, dL_dviewmatrix, dL_dquat, dL_dtrans, (hipStream_t)stream));
    return DGR_OK;
}
int dgr_l1_loss_scratch_floats(void) { return dgr::l1_loss_partials(); }
int dgr_l1_loss_forward(void* stream, long n_color, const float* color, const float* color_obs, long n_depth, const float* depth,
                        const float* depth_obs, float w_color, float w_depth, float* scratch, float* loss) {
    if (n_color < 0 || n_depth < 0 || !scratch || !loss || (n_color > 0 && (!color || !color_obs)) ||
        (n_depth > 0 && (!depth || !depth_obs))) {
        g_last_error = "dgr_l1_loss_forward: bad argument";
        return DGR_ERR_BAD_ARGUMENT;
    }
    HIP_TRY(dgr::launch_l1_loss_forward(n_color, color, color_obs, n_depth, depth, depth_obs, w_color, w_depth, scratch, loss,
                                        (hipStream_t)stream));
    return DGR_OK;
}
int dgr_l1_loss_backward(void* stream, long n_color, const float* color, const float* color_obs, long n_depth, const float* depth,
                         const float* depth_obs, float w_color, float w_depth, const float* upstream, float* dL_dcolor,
                         float* dL_ddepth) {
    if (n_color < 0 || n_depth < 0 || (n_color > 0 && (!color || !color_obs || !dL_dcolor)) ||
        (n_depth > 0 && (!depth || !depth_obs || !dL_ddepth))) {
        g_last_error = "dgr_l1_loss_backward: bad argument";
        return DGR_ERR_BAD_ARGUMENT;
    }
    HIP_TRY(dgr::launch_l1_loss_backward(n_color, color, color_obs, n_depth, depth, depth_obs, w_color, w_depth, upstream,
                                         dL_dcolor, dL_ddepth, (hipStream_t)stream));
    return DGR_OK;
}

// a free slot of the current device (g_status_mu held); creates one when all are busy
static long status_slot_acquire() {
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    for (size_t i = 0; i < g_status_slots.size(); i++) {
        StatusSlot& c = g_status_slots[i];
        if (c.busy || c.device != dev) continue;
        if (c.quarantined) {  // (given up by a poll: reusable once the stream its forward was queued on has drained)
            hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
            if (hipStreamIsCapturing(c.stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) continue;  // a query would invalidate the capture
            if (hipStreamQuery(c.stream) != hipSuccess) { (void)hipGetLastError(); continue; }
            c.quarantined = false;
        }
        return (long)i;
    }
    StatusSlot sl;
    HIP_TRY(hipEventCreateWithFlags(&sl.ev, hipEventDisableTiming));
    HIP_TRY(hipHostMalloc((void**)&sl.pinned, 8 * sizeof(int), hipHostMallocMapped | hipHostMallocCoherent));
    HIP_TRY(hipHostGetDevicePointer((void**)&sl.pinned_dev, sl.pinned, 0));
    HIP_TRY(hipMalloc((void**)&sl.ws, 16 * sizeof(uint32_t)));
    HIP_TRY(hipMemset(sl.ws, 0, 16 * sizeof(uint32_t)));  // (once per slot; the kernels keep the words zero between forwards)
    for (int i = 0; i < 8; i++) sl.pinned[i] = 0;
    sl.device = dev;
    g_status_slots.push_back(sl);
    return (long)g_status_slots.size() - 1;
}

long dgr_status_post(void* stream, const int* device_status) {
    if (!device_status) { g_last_error = "dgr_status_post: NULL"; return DGR_ERR_BAD_ARGUMENT; }
    std::lock_guard<std::mutex> lk(g_status_mu);
    const long id = status_slot_acquire();
    if (id < 0) return id;
    StatusSlot& sl = g_status_slots[(size_t)id];
    HIP_TRY(hipMemcpyAsync(sl.pinned, device_status, 4 * sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(hipEventRecord(sl.ev, (hipStream_t)stream));
    sl.busy = true;
    sl.mapped = false;
    return id;
}

long dgr_status_arm(void) {
    std::lock_guard<std::mutex> lk(g_status_mu);
    if (g_armed_slot >= 0) {  // armed twice without a forward in between: the first arm is withdrawn
        g_status_slots[(size_t)g_armed_slot].busy = false;
        g_armed_slot = -1;
    }
    const long id = status_slot_acquire();
    if (id < 0) return id;
    StatusSlot& sl = g_status_slots[(size_t)id];
    if (++g_status_tag == 0u) ++g_status_tag;
    sl.tag = g_status_tag;
    sl.busy = true;
    sl.mapped = true;
    ((volatile int*)sl.pinned)[4] = 0;
    g_armed_slot = id;
    return id;
}

int dgr_status_poll(long ticket, int wait, int* host_status4) {
    hipEvent_t ev;
    int* pinned;
    bool mapped, enqueued;
    uint32_t tag;
    hipStream_t stream;
    {
        std::lock_guard<std::mutex> lk(g_status_mu);
        if (ticket < 0 || (size_t)ticket >= g_status_slots.size() || !g_status_slots[(size_t)ticket].busy || !host_status4) {
            g_last_error = "dgr_status_poll: bad ticket";
            return DGR_ERR_BAD_ARGUMENT;
        }
        if (ticket == g_armed_slot) { g_last_error = "dgr_status_poll: the slot is armed and no forward has taken it"; return DGR_ERR_BAD_ARGUMENT; }
        ev = g_status_slots[(size_t)ticket].ev;
        pinned = g_status_slots[(size_t)ticket].pinned;
        mapped = g_status_slots[(size_t)ticket].mapped;
        tag = g_status_slots[(size_t)ticket].tag;
        stream = g_status_slots[(size_t)ticket].stream;
        enqueued = g_status_slots[(size_t)ticket].enqueued;
    }
    if (mapped) {  // written by the forward blend's first workgroup, the tag last: nothing to wait for on a stream
        const auto tag_here = [&] { return __atomic_load_n(pinned + 4, __ATOMIC_ACQUIRE) == (int)tag; };
        if (!tag_here()) {
            if (!wait) return 0;
            // Poll for a while, then stop burning the core (as wait_event_spinning).  The tag comes from ONE workgroup of ONE kernel:
            // if an earlier kernel of that forward faults, the device hangs or the stream was being captured when the forward was
            // issued, it never arrives -- so every millisecond the stream itself is asked: an error ends the wait with that error, a
            // stream that has finished all its work without the tag having been written ends it too, and so does a hard limit
            // (DGR_STATUS_TIMEOUT_MS, default 30 000).
            // (DGR_STATUS_TIMEOUT_MS = 0: no limit -- profiler replays and collectives' stragglers can legitimately hold a queue
            //  for longer than any default)
            static const long limit_ms = [] { const char* e = getenv("DGR_STATUS_TIMEOUT_MS"); return e ? (atol(e) > 0 ? atol(e) : 0L) : 30000L; }();
            const auto t0 = std::chrono::steady_clock::now();
            auto next_query = t0 + std::chrono::milliseconds(1);
            auto release = [&](const char* why, bool quarantine) {
                std::lock_guard<std::mutex> lk(g_status_mu);
                g_status_slots[(size_t)ticket].busy = false;
                g_status_slots[(size_t)ticket].quarantined = quarantine;  // the forward may still be queued and write the slot later
                g_last_error = why;
                return DGR_ERR_HIP;
            };
            while (!tag_here()) {
                const auto now = std::chrono::steady_clock::now();
                if (now - t0 > std::chrono::microseconds(400)) std::this_thread::sleep_for(std::chrono::microseconds(50));
                if (now < next_query) continue;
                next_query = now + std::chrono::milliseconds(1);
                if (enqueued) {
                    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
                    const bool capturing = hipStreamIsCapturing(stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
                    if (!capturing) {  // (a query on a capturing stream invalidates the capture)
                        const hipError_t e = hipStreamQuery(stream);
                        if (e == hipSuccess) {  // everything enqueued on the stream has completed: the tag is there, or it never will be
                            if (tag_here()) break;
                            return release("dgr_status_poll: the forward's stream is idle and its status word never arrived (was the forward "
                                           "issued while the stream was being captured?)", false);
                        }
                        if (e != hipErrorNotReady) {
                            (void)release("", true);
                            return hip_fail(e, "dgr_status_poll: hipStreamQuery on the forward's stream");
                        }
                    }
                }
                if (limit_ms > 0 && now - t0 > std::chrono::milliseconds(limit_ms))
                    return release("dgr_status_poll: timed out waiting for the forward's status word (DGR_STATUS_TIMEOUT_MS; 0 = no limit)", true);
            }
        }
        const volatile int* w = pinned;
        int word[8];
        for (int i = 0; i < 8; i++) word[i] = w[i];
        for (int i = 0; i < 4; i++) host_status4[i] = word[i];
        std::lock_guard<std::mutex> lk(g_status_mu);
        note_schedule_hint(g_status_slots[(size_t)ticket], word);
        g_status_slots[(size_t)ticket].busy = false;
        return 1;
    }
    if (wait) {
        HIP_TRY(wait_event_spinning(ev));
    } else {
        const hipError_t e = hipEventQuery(ev);
        if (e == hipErrorNotReady) return 0;
        if (e != hipSuccess) return hip_fail(e, "hipEventQuery");
    }
    for (int i = 0; i < 4; i++) host_status4[i] = pinned[i];
    std::lock_guard<std::mutex> lk(g_status_mu);
    g_status_slots[(size_t)ticket].busy = false;
    return 1;
}

int dgr_stream_is_capturing(void* stream) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing((hipStream_t)stream, &st) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return st == hipStreamCaptureStatusActive ? 1 : 0;
}

int dgr_backward_scratch_clean_arm(void) {
    g_scratch_clean_armed = true;
    return DGR_OK;
}

int dgr_early_status_arm(void) {
    g_early.armed = true;
    g_early.pending = false;
    return DGR_OK;
}
int dgr_early_status_wait(int* host_status4) {
    if (!host_status4) { g_last_error = "dgr_early_status_wait: NULL"; return DGR_ERR_BAD_ARGUMENT; }
    if (!g_early.pending) {  // nothing was posted (P == 0, or no presized forward since arming)
        g_early.armed = false;
        host_status4[0] = host_status4[1] = host_status4[2] = host_status4[3] = 0;
        return 1;
    }
    HIP_TRY(wait_event_spinning(g_early.ev));
    for (int i = 0; i < 4; i++) host_status4[i] = g_early.pinned[i];
    g_early.pending = false;
    return DGR_OK;
}

int dgr_set_option(const char* name, int value) {
    const std::string n(name ? name : "");
    if (n == "blend_wgs_per_cu") { g_blend_wgs_per_cu.store((value >= 3 && value <= 7) ? value : 0); return DGR_OK; }
    if (n == "tight_cull") { g_tight_cull.store(value ? 1 : 0); return DGR_OK; }
    if (n == "tile_schedule") { g_tile_schedule.store(value < 0 ? 0 : value > 2 ? 2 : value); return DGR_OK; }
    if (n == "fast_alpha") {  // (the option's name before alpha_mode 2 existed)
        g_alpha_mode.store(value ? 1 : 0);
        return DGR_OK;
    }
    if (n == "alpha_mode") {
        if (value < 0 || value > 2) { g_last_error = "alpha_mode: 0 (restatement's bits, fp32 expf), 1 (fast), 2 (glibc's expf form)"; return DGR_ERR_BAD_ARGUMENT; }
        g_alpha_mode.store(value);
        return DGR_OK;
    }
    if (n == "lds_count") { g_lds_count.store(value < 0 ? 0 : value > 2 ? 2 : value); return DGR_OK; }
    if (n == "lane_lists") { g_lane_lists.store(value < 0 ? 0 : value > 2 ? 2 : value); return DGR_OK; }
    if (n == "deterministic_grads") { g_det_grads.store(value ? 1 : 0); return DGR_OK; }
    if (n == "profile_every") { g_profile_every.store(value > 0 ? value : 1); return DGR_OK; }
    if (n == "batch_order") { g_batch_order.store(value ? 1 : 0); return DGR_OK; }
    if (n == "batch_streams") { g_batch_streams.store(value < 1 ? 1 : value > DGR_BATCH_MAX_STREAMS ? DGR_BATCH_MAX_STREAMS : value); return DGR_OK; }
    g_last_error = "unknown option: " + n;
    return DGR_ERR_BAD_ARGUMENT;
}
int dgr_get_option(const char* name) {
    const std::string n(name ? name : "");
    if (n == "blend_wgs_per_cu") return g_blend_wgs_per_cu.load();
    if (n == "tight_cull") return g_tight_cull.load();
    if (n == "tile_schedule") return g_tile_schedule.load();
    if (n == "fast_alpha") return g_alpha_mode.load() == 1 ? 1 : 0;
    if (n == "alpha_mode") return g_alpha_mode.load();
    if (n == "lds_count") return g_lds_count.load();
    if (n == "lane_lists") return g_lane_lists.load();
    if (n == "deterministic_grads") return g_det_grads.load();
    if (n == "profile_every") return g_profile_every.load();
    if (n == "batch_streams") return g_batch_streams.load();
    if (n == "batch_order") return g_batch_order.load();
    return DGR_ERR_BAD_ARGUMENT;
}

int dgr_set_thread_option(const char* name, int value) {
    const std::string n(name ? name : "");
    if (n == "alpha_mode") {
        if (value > 2) { g_last_error = "alpha_mode: 0, 1, 2 (or < 0: the process-wide option)"; return DGR_ERR_BAD_ARGUMENT; }
        t_alpha_mode = value < 0 ? -1 : value;
        return DGR_OK;
    }
    if (n == "fast_alpha") { t_alpha_mode = value < 0 ? -1 : (value ? 1 : 0); return DGR_OK; }
    if (n == "tight_cull") { t_tight_cull = value < 0 ? -1 : (value ? 1 : 0); return DGR_OK; }
    if (n == "deterministic_grads") { t_det_grads = value < 0 ? -1 : (value ? 1 : 0); return DGR_OK; }
    g_last_error = "not a per-thread option: " + n;
    return DGR_ERR_BAD_ARGUMENT;
}
int dgr_get_thread_option(const char* name) {
    const std::string n(name ? name : "");
    if (n == "alpha_mode") return opt_alpha_mode();
    if (n == "fast_alpha") return opt_alpha_mode() == 1 ? 1 : 0;
    if (n == "tight_cull") return opt_tight_cull();
    if (n == "deterministic_grads") return opt_det_grads();
    return DGR_ERR_BAD_ARGUMENT;
}
// the three as one word, each field = value + 1 (0 = "inherit", in an override word): bits 0-3 alpha_mode, 4-7 tight_cull, 8-11
// deterministic_grads
int dgr_thread_options_effective(void) { return (opt_alpha_mode() + 1) | ((opt_tight_cull() + 1) << 4) | ((opt_det_grads() + 1) << 8); }
int dgr_thread_options_swap(int word) {
    const int prev = (t_alpha_mode + 1) | ((t_tight_cull + 1) << 4) | ((t_det_grads + 1) << 8);
    if (word >= 0) {
        t_alpha_mode = (word & 15) - 1;
        t_tight_cull = ((word >> 4) & 15) - 1;
        t_det_grads = ((word >> 8) & 15) - 1;
    }
    return prev;
}

int dgr_profile_select(const char* stage) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    const std::string n(stage ? stage : "");
    bool found = n.empty() || n == "all";
    for (auto& p : g_prof) {
        p.seen = 0;
        p.on = (n == "all") || (n == p.name);
        found = found || p.on;
    }
    return found ? DGR_OK : DGR_ERR_BAD_ARGUMENT;
}
int dgr_profile_stage_count(void) { return ST_COUNT; }
const char* dgr_profile_stage_name(int i) { return (i >= 0 && i < ST_COUNT) ? g_prof[i].name : ""; }
int dgr_profile_read(const char* stage, double* total_ms, int* launches) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& p : g_prof) {
        if (std::string(stage) != p.name) continue;
        double tot = 0;
        int n = 0;
        for (auto& e : p.ev) {
            float ms = 0;
            if (hipEventSynchronize(e.second) == hipSuccess && hipEventElapsedTime(&ms, e.first, e.second) == hipSuccess) {
                tot += ms;
                n++;
            }
            (void)hipEventDestroy(e.first);
            (void)hipEventDestroy(e.second);
        }
        p.ev.clear();
        *total_ms = tot;
        *launches = n;
        return DGR_OK;
    }
    return DGR_ERR_BAD_ARGUMENT;
}

long dgr_state_export(void* stream, const char* name, int P, int width, int height, int num_rendered,
                      int binning_capacity, const char* geom_buffer, const char* binning_buffer, const char* image_buffer,
                      void* dst) {
    hipStream_t st = (hipStream_t)stream;
    if (binning_capacity < num_rendered) { g_last_error = "binning_capacity < num_rendered"; return -1; }
    dgr::GeometryView g = dgr::carve_geometry(const_cast<char*>(geom_buffer), P);
    dgr::ImageView img = dgr::carve_image(const_cast<char*>(image_buffer), width, height);
    dgr::BinningView bin = dgr::carve_binning(const_cast<char*>(binning_buffer), (size_t)binning_capacity);
    const size_t tiles = (size_t)dgr::tiles_x(width) * dgr::tiles_y(height), N = (size_t)width * height;
    auto copy = [&](const void* src, size_t bytes) -> int {
        HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st));
        return 0;
    };
    auto geomk = [&](int kind) -> int {
        if (P > 0) hipLaunchKernelGGL(export_geom_kernel, dim3((P + 255) / 256), dim3(256), 0, st, kind, P, g, dst);
        HIP_TRY(hipGetLastError());
        return 0;
    };
    const std::string n(name);
    if (n == "depths") return copy(g.depths, 4 * (size_t)P) ? -1 : P;
    if (n == "radii") return copy(g.radii, 4 * (size_t)P) ? -1 : P;
    if (n == "means2D") return geomk(EX_MEANS2D) ? -1 : 2L * P;
    if (n == "conic_opacity") return geomk(EX_CONIC_OPACITY) ? -1 : 4L * P;
    if (n == "rgb") return geomk(EX_RGB) ? -1 : 3L * P;
    if (n == "clamped") return geomk(EX_CLAMPED) ? -1 : 3L * P;
    if (n == "tiles_touched") return geomk(EX_TILES_TOUCHED) ? -1 : P;
    if (n == "point_list" || n == "contribution_tags" || n == "half_tags") {
        if (num_rendered > 0) {
            const dim3 grid((num_rendered + 255) / 256);
            if (n == "point_list")
                hipLaunchKernelGGL(export_point_list_kernel, grid, dim3(256), 0, st, bin.point_list, (uint32_t*)dst, num_rendered);
            else  // (the tag bytes: in the binning's pair_cov bytes, render_common.h)
                hipLaunchKernelGGL(export_tag_bytes_kernel, grid, dim3(256), 0, st, bin.pair_cov, (uint8_t*)dst, num_rendered, n == "half_tags" ? 1 : 0);
            if (hipGetLastError() != hipSuccess) return -1;
        }
        return num_rendered;
    }
    if (n == "keys") {
        hipLaunchKernelGGL(export_keys_kernel, dim3((unsigned)tiles), dim3(256), 0, st, img, bin, g, (uint64_t*)dst);
        if (hipGetLastError() != hipSuccess) return -1;
        return num_rendered;
    }
    if (n == "ranges") return copy(img.ranges, 8 * tiles) ? -1 : (long)(2 * tiles);
    if (n == "tile_sched") return copy(img.tile_sched, 16 * tiles) ? -1 : (long)(4 * tiles);
    if (n == "sched_flag") return copy(img.cursor + 3, 4) ? -1 : 1L;  // 1: this frame's blend kernels walk tile_sched, 0: the static band map
    if (n == "n_contrib") return copy(img.n_contrib, 4 * N) ? -1 : (long)N;
    if (n == "n_valid") return copy(img.n_valid, 4 * N) ? -1 : (long)N;
    if (n == "final_T") return copy(img.final_T, 4 * N) ? -1 : (long)N;
    g_last_error = "unknown state array: " + n;
    return -1;
}

}  // extern "C"
