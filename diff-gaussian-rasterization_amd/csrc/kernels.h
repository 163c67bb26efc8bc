// kernels.h -- argument blocks and launch entry points of the gfx950 kernels (internal).
#pragma once
#include <hip/hip_ext.h>
#include "dgr_common.h"

namespace dgr {

// Kernel launches go through dgr::launch().  When the profiler (dgr_profile_*) brackets a stage it sets
// `g_launch_events` for the calling thread and the next kernel is launched with hipExtLaunchKernelGGL, which puts
// the two events into the kernel's own dispatch packet: they hold the kernel's start and end, the same
// timestamps rocprofv3 reports, also when other streams keep the GPU busy (a hipEventRecord bracket would include
// the time the kernel's waves wait for compute units).
struct LaunchEvents {
    hipEvent_t start, stop;
    bool used;
};
extern thread_local LaunchEvents* g_launch_events;  // api.hip

template <typename K, typename... Args>
inline void launch(K kernel, dim3 grid, dim3 block, hipStream_t stream, Args... args) {
    LaunchEvents* ev = g_launch_events;
    if (ev && !ev->used) {
        ev->used = true;
        hipExtLaunchKernelGGL(kernel, grid, block, 0, stream, ev->start, ev->stop, 0, args...);
    } else {
        hipLaunchKernelGGL(kernel, grid, block, 0, stream, args...);
    }
}
template <typename K, typename... Args>
inline void launch_shmem(K kernel, dim3 grid, dim3 block, size_t shmem, hipStream_t stream, Args... args) {
    LaunchEvents* ev = g_launch_events;
    if (ev && !ev->used) {
        ev->used = true;
        hipExtLaunchKernelGGL(kernel, grid, block, shmem, stream, ev->start, ev->stop, 0, args...);
    } else {
        hipLaunchKernelGGL(kernel, grid, block, shmem, stream, args...);
    }
}

// Blend kernels, when several views are in flight: they run at 8 workgroups per CU, i.e. they hold every wave slot, the
// whole register file and 144 of the 160 KB of LDS, so another stream's kernels enter a CU only as blend workgroups drain.
// dgr_set_option("blend_wgs_per_cu", n) (3 <= n <= 7; 0 = no cap) pads each blend workgroup with unused dynamic LDS so
// that at most n fit a CU, which leaves wave slots, registers and LDS for the other views' bandwidth- and latency-bound
// kernels.  blend_pad_bytes() (api.hip) caches the kernels' static LDS sizes.
size_t blend_pad_bytes(const void* kernel);
template <typename K, typename... Args>
inline void launch_blend(K kernel, dim3 grid, dim3 block, hipStream_t stream, Args... args) {
    launch_shmem(kernel, grid, block, blend_pad_bytes(reinterpret_cast<const void*>(kernel)), stream, args...);
}

struct PreprocessFwdArgs {
    int P, D, M, W, H, grid_x, grid_y;
    const float* means3D;
    const float* scales;
    float scale_modifier;
    const float* rotations;
    const float* opacities;
    const float* shs;
    const float* cov3D_precomp;
    const float* colors_precomp;
    const float* view;
    const float* proj;
    const float* campos;
    float tan_fovx, tan_fovy, focal_x, focal_y;
    int prefiltered;
    int tight_cull;        // dgr_set_option("tight_cull"): alpha-aware tile rectangles (changes the integer path)
    bool sh_vec_ok;
    GeometryView geom;
    int* radii_out;        // caller's radii tensor (may be NULL)
    uint32_t* zero_words;  // n_zero_words 32-bit words to clear (the padded tile counters)
    int n_zero_words;
    float* gau_uncertainty;   // [P] cleared here, accumulated by the forward blend (may be NULL)
    int* gau_related_pixels;  // [P] likewise
    // presized path (binning buffer exists before the kernel runs): the block also takes the per-instance tile-counter
    // atomics and stores the arrival ranks -- count_rank.h; offsets come from a global cursor, not from a scan over P.
    // The counters and the cursor must be zero when the kernel starts (launch_zero_fill before it).
    int fused_count;
    uint32_t* tile_count;
    uint32_t* cursor;
    uint32_t* ranks;
    int capacity;
};

// ---- batched views: what differs between the views of a batch; everything else is in `base`
#ifndef DGR_MAX_BATCH_VIEWS
#define DGR_MAX_BATCH_VIEWS 8  // (also in include/dgr_hip.h)
#endif
struct FwdViewPart {
    const float* view;
    const float* proj;
    const float* campos;
    GeometryView geom;
    int* radii_out;
    float* gau_uncertainty;
    int* gau_related_pixels;
};
struct PreprocessFwdBatchArgs {
    PreprocessFwdArgs base;  // view / proj / campos / geom / radii_out / gau_* unused
    int V;
    FwdViewPart v[DGR_MAX_BATCH_VIEWS];
};

struct PreprocessBwdArgs {
    int P, D, M, W, H;
    const float* means3D;
    const int* radii;
    const float* shs;
    const float* scales;
    const float* rotations;
    float scale_modifier;
    const float* cov3D_precomp;
    const float* view;
    const float* proj;
    const float* campos;
    const float* perspec;
    float tan_fovx, tan_fovy, focal_x, focal_y;
    bool sh_vec_ok;
    int track_off, map_off;
    int full_variant;   // 1: semantics of F/cuda_rasterizer/backward.cu (computeCov2DCUDA assigns + depth term, pose from
                        //    dgc_dCampos / colour-only ndc sums / front-most depth sums); 0: light
    GeometryView geom;
    float* acc;         // [P,16] sums written by the blend backward
    double* det_pose;   // deterministic gradients: [blocks, 12] per-block pose partials (NULL: the 64 bucket rows, double atomics)
    int clear_scratch;  // 1: leave the scratch as it was found -- all zero: every accumulator row read is cleared by its reader and
                        //    the block that finishes the pose sum clears the buckets and the ticket (dgr_backward_scratch_clean_arm)
    float* dL_dmean2D;  // [P,3]
    float* dL_dconic;   // [P,4] optional
    float* dL_dopacity; // [P]
    float* dL_dcolor;   // [P,3] optional
    float* dL_ddepth;   // [P]   optional
    float* dL_dmean3D;  // [P,3]
    float* dL_dcov3D;   // [P,6]
    float* dL_dsh;      // [P,M,3] optional (M == 0)
    float* dL_dscale;   // [P,3]
    float* dL_drot;     // [P,4]
    double* pose_part;  // [DGR_POSE_BUCKETS,12], zero on entry
    uint32_t* ticket;   // zero on entry
    float* dL_dview;    // [16] written by the last block to deliver its pose partial
};

struct BwdViewPart {
    const float* view;
    const float* proj;
    const float* campos;
    const float* perspec;
    const int* radii;
    GeometryView geom;
    const float* acc;     // this view's accumulator rows (its backward scratch)
    float* dL_dmean2D;    // [P,3] per view (densification statistics are per view); may be NULL
    double* pose_part;
    uint32_t* ticket;
    float* dL_dview;      // [16]
    double* det_pose;     // deterministic gradients: this view's [blocks, 12] per-block pose partials (NULL otherwise; all views alike)
};
struct PreprocessBwdBatchArgs {
    PreprocessBwdArgs base;  // the per-view members unused; the dense outputs receive the SUM over the views
    int V;
    BwdViewPart v[DGR_MAX_BATCH_VIEWS];
};

struct RenderFwdLightArgs {
    int W, H, grid_x, grid_y;
    const uint4* sched;    // [tiles] {tile, list start, list end, -} of workgroup b (ImageView::tile_sched)
    const uint2* ranges;   // [tiles] ... and without a schedule (ImageView::cursor[3] == 0): block -> tile by the static XCD band
    const uint32_t* sched_flag;  // map, list from the range table (render_common.h: blend_slot)
    uint32_t* point_list;  // read only (the light forward keeps its contribution tags in the byte array beside it: render_common.h, half_tags)
    const float4* rec;
    const float* bg;
    const float* gt_depth;
    float* out_color;
    float* out_depth;
    float* out_median;
    float* out_alpha;
    float* out_depth_var;
    uint32_t* n_contrib;
    float* gau_uncertainty;
    int* gau_related_pixels;
    StatusReport rep;      // armed status slot (dgr_status_arm): workgroup 0 copies the frame's status word to the host
    const int* status;
};

struct RenderBwdLightArgs {
    int W, H, grid_x, grid_y;
    const uint4* sched;    // [tiles] {tile, list start, list end, -} of workgroup b (ImageView::tile_sched)
    const uint2* ranges;   // [tiles] ... and without a schedule (ImageView::cursor[3] == 0): block -> tile by the static XCD band
    const uint32_t* sched_flag;  // map, list from the range table (render_common.h: blend_slot)
    const uint32_t* point_list;
    const float4* rec;
    const float* bg;
    const float* gt_depth;
    const float* alphas;
    const uint32_t* n_contrib;
    const float* dL_dpix;
    const float* dL_dpix_depth;
    const float* dL_dpix_median;
    const float* dL_dpix_var;
    const float* means3D;
    const float* view;
    float* acc;  // [P,16]
    int track_off, map_off;
    // deterministic gradients (render_light.hip: DET): the instance-major row buffer [R,16] (NULL: off), the Gaussians' tile
    // rectangles and first-instance offsets
    float* det_rows;
    const ushort4* det_rect;
    const uint32_t* det_goff;
    uint32_t det_R;  // rows of det_rows (>= the frame's instances; a row index beyond it is dropped, not written)
};

struct RenderFwdFullArgs {
    int W, H, grid_x, grid_y;
    const uint4* sched;    // [tiles] {tile, list start, list end, -} of workgroup b (ImageView::tile_sched)
    const uint2* ranges;   // [tiles] ... and without a schedule (ImageView::cursor[3] == 0): block -> tile by the static XCD band
    const uint32_t* sched_flag;  // map, list from the range table (render_common.h: blend_slot)
    uint32_t* point_list;  // read only (the contribution tags go into the byte array beside it: render_common.h, half_tags)
    const float4* rec;
    const float* bg;
    float* out_color;
    float* out_depth;
    float* out_uncertainty;
    uint32_t* n_contrib;
    uint32_t* n_valid;
    uint32_t* first_contrib;
    float* final_T;
    int* status;  // status[3] += sum of n_valid (num_related_primitives)
    StatusReport rep;  // armed status slot (dgr_status_arm): workgroup 0 copies the frame's status word to the host
};

struct RenderBwdFullArgs {
    int W, H, grid_x, grid_y;
    const uint4* sched;    // [tiles] {tile, list start, list end, -} of workgroup b (ImageView::tile_sched)
    const uint2* ranges;   // [tiles] ... and without a schedule (ImageView::cursor[3] == 0): block -> tile by the static XCD band
    const uint32_t* sched_flag;  // map, list from the range table (render_common.h: blend_slot)
    const uint32_t* point_list;
    const float4* rec;
    const float* bg;
    const float* gt_depth;
    const float* final_T;
    const uint32_t* n_contrib;
    const uint32_t* first_contrib;
    const float* dL_dpix;
    const float* dL_depths;
    const float* dL_duncertainties;
    float* acc;  // [P,16]
    // deterministic gradients (render_light.hip: DET): the finished row of a (tile, Gaussian) pair is stored to det_rows instead
    float* det_rows;
    const ushort4* det_rect;
    const uint32_t* det_goff;
    uint32_t det_R;
};

// ---- launchers (each enqueues on `stream` and returns the hipError_t of the launch) ----
hipError_t launch_preprocess_fwd(const PreprocessFwdArgs& a, hipStream_t stream);
hipError_t launch_preprocess_bwd(const PreprocessBwdArgs& a, hipStream_t stream);
hipError_t launch_preprocess_fwd_batch(const PreprocessFwdBatchArgs& b, hipStream_t stream);
hipError_t launch_preprocess_bwd_batch(const PreprocessBwdBatchArgs& b, hipStream_t stream);
// zero-fill of a 16-byte aligned buffer whose size is a multiple of 16 (a kernel rather than hipMemsetAsync: memset nodes of a
// captured hipGraph were seen to re-execute with corrupted parameters on this ROCm)
hipError_t launch_zero_fill(void* dst, size_t bytes, hipStream_t stream);
// fused sparse Adam (optim.hip): rows with visible[row] <= 0 are skipped entirely; visible == NULL updates every row
// slam.hip
hipError_t launch_pose_forward(const float* q, const float* t, const float* perspec, float* view, float* proj, float* campos,
                               hipStream_t stream);
hipError_t launch_pose_backward(const float* q, const float* dview, float* dq, float* dt, hipStream_t stream);
int l1_loss_partials();
hipError_t launch_l1_loss_forward(long n_c, const float* c, const float* c_obs, long n_d, const float* d, const float* d_obs,
                                  float w_c, float w_d, float* partial, float* loss, hipStream_t stream);
hipError_t launch_l1_loss_backward(long n_c, const float* c, const float* c_obs, long n_d, const float* d, const float* d_obs,
                                   float w_c, float w_d, const float* upstream, float* dc, float* dd, hipStream_t stream);
hipError_t launch_densification_stats(int rows, const float* dmeans2D, const int* radii, float* grad_accum, float* denom,
                                      float* max_radii2D, hipStream_t stream);
hipError_t launch_sparse_adam(size_t rows, int k, float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                              const int* visible, float lr, float beta1, float beta2, float eps, int step,
                              const int* step_dev, hipStream_t stream);
// view-independent covariance of a batch of views (preprocess.hip)
hipError_t launch_cov3d_forward(int P, const float* scales, const float* rotations, float mod, float* cov3D, hipStream_t stream);
hipError_t launch_cov3d_backward(int P, const float* scales, const float* rotations, float mod, const float* dL_dcov3D,
                                 float* dL_dscale, float* dL_drot, hipStream_t stream);
hipError_t launch_mark_visible(int P, const float* means, const float* view, uint8_t* present, hipStream_t stream);

// binning: tile_count -> ranges (+ total in status[0], overflow in status[1]); emit keys; sort tiles
// `fused`: preprocess_fwd counted (no scan_blocks ran): scan_tiles then also fills status[2] (from the cursor's violation
// word) and status[3]
hipError_t launch_scan_tiles(ImageView img, int tiles, int grid_x, int capacity, bool fused, int blend_flags, StatusReport rep,
                             hipStream_t stream);
hipError_t launch_count_rank(int P, GeometryView geom, ImageView img, BinningView bin, int grid_x, int capacity,
                             hipStream_t stream);
hipError_t launch_scan_blocks(int P, GeometryView geom, ImageView img, hipStream_t stream);
hipError_t launch_emit_instances(int P, GeometryView geom, ImageView img, BinningView bin, int grid_x, hipStream_t stream);
// two-level binning (segment_binning.hip): presized and callback paths, frames whose segment tables fit LDS.
// `prefixed`: geom.block_tiles is already the exclusive prefix of the block totals and the status word is initialised
// (callback path, after scan_blocks)
bool segment_binning_fits(int W, int H);
int segment_binning_workgroups(int P);
int segment_shift(int W, int H, int capacity, int longest_list = -1);  // log2 of the tiles per segment (4, 3 or 2) for this frame, capacity and (if known) longest list
hipError_t launch_bin_segments(int P, GeometryView geom, BinningView bin, SegmentTables tb, int grid_x, int grid_y, int seg_shift,
                               int capacity, bool prefixed, hipStream_t stream);
extern unsigned long long* g_bin_tiles_trace;  // debug: phase time stamps per bin_tiles workgroup (segment_binning.hip)
hipError_t launch_bin_tiles(int P, GeometryView geom, ImageView img, BinningView bin, SegmentTables tb, int grid_x, int grid_y,
                            int seg_shift, int capacity, bool prefixed, int blend_flags, StatusReport rep, hipStream_t stream);
hipError_t launch_sort_tiles(ImageView img, BinningView bin, int tiles, hipStream_t stream);
// ranges -> the blend kernels' schedule (img.tile_sched): tiles by descending list length, so that the longest lists
// start first and every XCD gets its share of a cluster
hipError_t launch_tile_schedule(ImageView img, int tiles, hipStream_t stream);

// alpha_mode: render_common.h (0 = ALPHA_REF, the restatement's bits; 1 = ALPHA_FAST; 2 = ALPHA_GLIBC)
hipError_t launch_render_fwd_light(const RenderFwdLightArgs& a, int alpha_mode, hipStream_t stream);
hipError_t launch_render_bwd_light(const RenderBwdLightArgs& a, int alpha_mode, hipStream_t stream);
hipError_t launch_render_fwd_full(const RenderFwdFullArgs& a, int alpha_mode, hipStream_t stream);
hipError_t launch_render_bwd_full(const RenderBwdFullArgs& a, int alpha_mode, hipStream_t stream, bool deterministic = false);
hipError_t launch_det_offsets(int P, const ushort4* rect, uint32_t* blk, uint32_t* goff, hipStream_t stream);
hipError_t launch_det_gather(int P, const ushort4* rect, const uint32_t* goff, const float* rows, uint32_t R, float* acc, hipStream_t stream);
hipError_t launch_half_reduce16_test(const float* in, float* r0, float* r1, int* slot0, int* slot1, hipStream_t stream);
hipError_t launch_half_reduce_test(const float* in, float* r0, float* r1, float* h3, int* slot0, int* slot1, int* comp3, hipStream_t stream);
hipError_t launch_lane_lists_test(const unsigned char* codes, uint32_t* paired, uint32_t* halves, hipStream_t stream);
hipError_t launch_wave_reduce_test(const float* in, float* out16, float* out12, float* out4, int* comp16, int* comp12, int* comp4,
                                   hipStream_t stream);
hipError_t launch_exact_math_test(int n, const float* x, const float* a, const float* b, float* out_exp, float* out_div, int alpha_mode,
                                  hipStream_t stream);

}  // namespace dgr
