/* dgr_hip.h -- C ABI of the MI355X (gfx950) differentiable Gaussian rasterizer.
 *
 * This is the drop-in boundary for the reference's hot path.  Each entry point replaces one
 * member of `CudaRasterizer::Rasterizer` (the raw-pointer layer the reference's torch binding
 * calls) and keeps its argument order and meaning; what changes is stated per function.
 * Reference paths: L = diff-gaussian-rasterization-light, F = diff-gaussian-rasterization-full.
 *
 * Conventions
 *  - every `float*` / `int*` / `char*` argument is a DEVICE pointer unless marked "host";
 *    optional inputs are NULL exactly where the reference passes nullptr (empty tensors);
 *  - `stream` is a hipStream_t passed as void* (NULL = the legacy default stream the reference uses);
 *  - 4x4 matrices are 16 floats read column-major: x' = m[0]x + m[4]y + m[8]z + m[12]
 *    (cuda_rasterizer/auxiliary.h:58-77);
 *  - the three state buffers (geometry / binning / image) are opaque to the caller, as in the
 *    reference (L/cuda_rasterizer/rasterizer_impl.h:29-64); their layout here is different and
 *    documented in DESIGN.md;
 *  - return value: >= 0 success (forward: num_rendered), < 0 one of DGR_ERR_*.
 */
#ifndef DGR_HIP_H
#define DGR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DGR_OK 0
#define DGR_ERR_BAD_ARGUMENT (-1)       /* e.g. non-RGB without precomputed colours (L/cr/rasterizer_impl.cu:248-251) */
#define DGR_ERR_PREFILTERED (-2)        /* a point was culled although `prefiltered` is set (cr/auxiliary.h:154-161: __trap there) */
#define DGR_ERR_BINNING_OVERFLOW (-3)   /* presized binning buffer too small; *num_rendered holds the required count */
#define DGR_ERR_HIP (-4)                /* a HIP runtime call failed; see dgr_last_error() */
#define DGR_ERR_ALLOC (-5)              /* an allocation callback returned NULL */

/* Replaces std::function<char*(size_t)> (L/cr/rasterizer.h:41-43, L/rasterize_points.cu:27-33):
 * called with the number of bytes needed, returns a device pointer (>= 256-byte aligned). */
typedef char* (*dgr_alloc_fn)(size_t bytes, void* user);

/* Human-readable text of the last DGR_ERR_HIP on this thread (host string, never NULL). */
const char* dgr_last_error(void);

/* Version / target string of the built library, e.g. "dgr_hip 0.1 gfx950". */
const char* dgr_version(void);

/* ---- state buffer sizes (the reference's `required<T>(n)`, L/cr/rasterizer_impl.h:66-72) ---- */
size_t dgr_geometry_bytes(int P);
size_t dgr_image_bytes(int width, int height);
/* 24 bytes per instance of capacity (sorted list, key scratch, arrival ranks / pair columns, pair keys) plus the segment
 * binning's forward-only tables (csrc/segment_binning.hip: 2 KB per 16-tile row segment), i.e. NON-ZERO for a capacity of 0:
 * the presized entry points need a binning buffer of this size for every P > 0 and reject NULL. */
size_t dgr_binning_bytes(int num_rendered_capacity, int width, int height);
/* scratch the backward needs (per-Gaussian accumulator rows + reduction partials) */
size_t dgr_light_backward_scratch_bytes(int P, int width, int height);
/* ... with the option "deterministic_grads" on: the scratch also holds the instance-major row buffer, 64 bytes per tile
 * instance -- R = the value the backward will be given as `R`, at least the forward's num_rendered (a lazy caller passes its
 * binning capacity).  Equals dgr_light_backward_scratch_bytes(P, W, H) while the option is off. */
size_t dgr_light_backward_scratch_bytes_r(int P, int W, int H, int R);

/* Replaces CudaRasterizer::Rasterizer::markVisible (L/cr/rasterizer.h:33-38,
 * L/cr/rasterizer_impl.cu:54-66,141-153).  present[i] = (view * p_i).z > 0.2.  `present` is P bytes (bool). */
int dgr_mark_visible(void* stream, int P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                     uint8_t* present);

/* Replaces CudaRasterizer::Rasterizer::forward of the light variant (L/cr/rasterizer.h:40-70,
 * L/cr/rasterizer_impl.cu:197-350).  Same arguments in the same order, with `stream` prepended and
 * the three std::function allocators replaced by C callbacks sharing one `alloc_user`.
 * Like the reference it blocks the host once (to size the binning buffer) and returns num_rendered.
 * Outputs need NOT be zero-initialised (the reference's binding zero-fills them first,
 * L/rasterize_points.cu:69-76; here every element is written).  `out_depth_var` is written as 0
 * (L/cr/forward.cu:317,410).  `radii` may be NULL (L/cr/rasterizer_impl.cu:235-238). */
int dgr_light_forward(void* stream, dgr_alloc_fn geometryBuffer, dgr_alloc_fn binningBuffer, dgr_alloc_fn imageBuffer,
                      void* alloc_user, int P, int D, int M, const float* background, int width, int height,
                      const float* means3D, const float* shs, const float* colors_precomp, const float* opacities,
                      const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
                      const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx,
                      float tan_fovy, int prefiltered, float* out_color, float* out_depth, float* out_median_depth,
                      float* out_alpha, const float* gt_depth, float* out_depth_var, float* gau_uncertainty,
                      int* gau_related_pixels, int* radii, int debug);

/* Same computation with caller-provided state buffers and NO host synchronisation (hipGraph-capturable):
 * `binning_buffer` holds at most `binning_capacity` instances.  `status` is a device int[4] written by the
 * kernels: {num_rendered, overflow flag, prefiltered-violation flag, reserved}.  When the capacity is too
 * small the blend kernels see empty tile lists and the caller must re-run with a larger buffer. */
int dgr_light_forward_presized(void* stream, char* geometry_buffer, char* binning_buffer, int binning_capacity,
                               char* image_buffer, int* status, int P, int D, int M, const float* background,
                               int width, int height, const float* means3D, const float* shs,
                               const float* colors_precomp, const float* opacities, const float* scales,
                               float scale_modifier, const float* rotations, const float* cov3D_precomp,
                               const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                               float tan_fovx, float tan_fovy, int prefiltered, float* out_color, float* out_depth,
                               float* out_median_depth, float* out_alpha, const float* gt_depth,
                               float* out_depth_var, float* gau_uncertainty, int* gau_related_pixels, int* radii);

/* Replaces CudaRasterizer::Rasterizer::backward of the light variant (L/cr/rasterizer.h:72-104,
 * L/cr/rasterizer_impl.cu:354-495).  Same arguments in the same order with `stream` prepended and a
 * scratch buffer appended.  Differences, all on buffers the reference's Python never sees:
 *  - `dL_dview` receives the 16 reduced entries (entries 3,7,11,15 = 0), i.e. what
 *    L/diff_gaussian_rasterization/__init__.py:160-161 obtains by summing the reference's [H*W,4,4] buffer;
 *  - `dgndcs_dviewmatrix` / `dg_camd_dviewmatrix` (per-Gaussian pose Jacobians, L/cr/backward.cu:701-751)
 *    are not materialised and may be NULL;
 *  - `dL_dconic` ([P,2,2]) and `dL_ddepth` ([P,1]) may be NULL; when given they receive the same sums
 *    the reference accumulates there;
 *  - gradient outputs need not be zero-initialised: rows of invisible Gaussians are written as 0;
 *  - every per-Gaussian gradient output may be NULL and is then not written.  A tracking step (only `viewmatrix` requires
 *    a gradient) passes them all as NULL together with map_off = 1: the backward then forms the pose gradient alone and
 *    moves no dense per-Gaussian rows (248 bytes per Gaussian at SH degree 3);
 *  - `dL_dpix_median_depth` and `dL_dpix_depth_var` may each be NULL (the loss did not use that output: an all-zero image).
 *    With both NULL the blend backward runs a leaner kernel -- no variance term, no median test: 8 % fewer issue cycles per
 *    list entry, bit-identical to all-zero images -- which is what the compiled autograd node passes when autograd hands it
 *    no gradient for those outputs (CG-SLAM's losses use neither: depth_var is identically zero, L/cr/forward.cu:317,410).
 * `R` is the value forward returned; `radii` may be NULL (internal copy is used).
 * The forward does not keep the 3D covariance: unless `cov3D_precomp` is given, `scales`, `rotations` and `scale_modifier`
 * must be the forward's (the backward re-forms the covariance from them, same expression, same bits) and may not be NULL
 * (DGR_ERR_BAD_ARGUMENT); likewise the three state buffers. */
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
                       const float* gt_depth, int track_off, int map_off, char* scratch, size_t scratch_bytes);

/* Resident scratch.  dgr_light_backward / dgr_full_backward clear `scratch` with one launch before the blend backward.  A caller
 * that keeps a scratch buffer across calls (one per stream: calls that share one must be ordered on a stream) can skip that launch:
 * after dgr_backward_scratch_clean_arm() the NEXT backward call of this thread takes `scratch` as ALL ZERO on entry and leaves it
 * all zero when its kernels have completed -- the per-Gaussian kernel clears every accumulator row it reads, the block that
 * finishes the pose sum clears the partials.  Start from a zero-filled buffer; after a call that returned an error, zero it again. */
int dgr_backward_scratch_clean_arm(void);

/* ---- -full variant -------------------------------------------------------------------------------
 * Replaces CudaRasterizer::Rasterizer::forward of the full variant (F/cr/rasterizer.h:31-62,
 * F/cr/rasterizer_impl.cu:349-500): returns num_rendered; the tuple's second member (num_related_primitives,
 * the total of n_valid_contrib) is written to *num_related_primitives (host int, may be NULL to skip the second
 * blocking read).  `out_uncertainty` is sum alpha*T (F/cr/forward.cu:367,394); gt_depth is accepted and unused in
 * the forward, as there (quirk F6). */
int dgr_full_forward(void* stream, dgr_alloc_fn geometryBuffer, dgr_alloc_fn binningBuffer, dgr_alloc_fn imageBuffer,
                     void* alloc_user, int P, int D, int M, const float* background, int width, int height,
                     const float* means3D, const float* shs, const float* colors_precomp, const float* opacities,
                     const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
                     const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx,
                     float tan_fovy, int prefiltered, float* out_color, float* out_depth, const float* gt_depth,
                     float* out_uncertainty, int* radii, int* num_related_primitives);

/* No-sync form; status = device int[4] {num_rendered, overflow, prefiltered violation, num_related_primitives}. */
int dgr_full_forward_presized(void* stream, char* geometry_buffer, char* binning_buffer, int binning_capacity,
                              char* image_buffer, int* status, int P, int D, int M, const float* background, int width,
                              int height, const float* means3D, const float* shs, const float* colors_precomp,
                              const float* opacities, const float* scales, float scale_modifier, const float* rotations,
                              const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                              const float* cam_pos, float tan_fovx, float tan_fovy, int prefiltered, float* out_color,
                              float* out_depth, const float* gt_depth, float* out_uncertainty, int* radii);

/* Replaces CudaRasterizer::Rasterizer::backward of the full variant (F/cr/rasterizer.h:64-102,
 * F/cr/rasterizer_impl.cu:504-666), same arguments in the same order (+ stream, + scratch).  The NG-sized pair
 * lists (dpixel_dgc, gau_id_list, pix_id_list, dpixel_dndcs, dpixel_dinvcovs, ddepth_dndcs, ddepth_dinvcovs) and
 * the per-Gaussian Jacobian tables (dgc_dCam_position, dgndcs_dviewmatrix, dgc_invcovs_dT) are implementation
 * scratch of the reference; they are not materialised here and may be NULL.  dL_dview receives the 4x4 gradient
 * with the well-defined semantics of ComputePG: every recorded (pixel, Gaussian) pair is consumed.  (The reference
 * lets threads of pixels without a valid contributor return before the block-wide loads, F/cr/backward.cu:875-878,
 * 935-938, which makes its own result undefined for the other pixels of such a tile; DESIGN.md "full variant".)
 * dL_duncertainties may be NULL (the loss did not use the uncertainty output): it reads as zero, and the blend backward
 * drops the variance recurrence (bit-identical to an all-zero image).  R: as in dgr_light_backward; read only with
 * "deterministic_grads", where it sizes the instance-major row buffer (scratch: dgr_light_backward_scratch_bytes_r). */
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
                      size_t scratch_bytes);

/* ---- stage-wise access for tests and profiling (views into the opaque state buffers) ---- */
/* Copies one named array of a state buffer to `dst` (device or host pointer).  Names: "depths", "radii",
 * "means2D", "conic_opacity", "rgb", "clamped", "tiles_touched" (geometry); "point_list", "keys" (binning); "contribution_tags"
 * (one byte per list entry, bit w = some pixel of quadrant w of the tile blended it: the forward blend's tag bytes folded),
 * "half_tags" (those bytes as they are: bit 2 w + h = half h of quadrant w);
 * "ranges", "tile_sched", "sched_flag" (one word, the frame's blend flags: bit 0 = its blend kernels use the schedule, bit 1 = the
 * binning buffer overflowed, bit 2 = quadrant lane lists: option "lane_lists"), "n_contrib", "n_valid",
 * "final_T" (image; the last two: full variant).  Layout conversion to the reference's element types is done on the fly.
 * `num_rendered` instances are exported; `binning_capacity` is the capacity the binning buffer was carved with
 * (= num_rendered after dgr_*_forward, the caller's capacity after *_presized).  Returns the element count, or < 0. */
long dgr_state_export(void* stream, const char* name, int P, int width, int height, int num_rendered,
                      int binning_capacity, const char* geom_buffer, const char* binning_buffer,
                      const char* image_buffer, void* dst_device);

/* Early status.  The reference blocks the host once per forward to read num_rendered (L/cr/rasterizer_impl.cu:287).  That
 * number and the prefiltered-violation flag are final after the per-block scan, about a tenth of the way into the forward:
 * dgr_early_status_arm() makes the NEXT *_forward_presized call of this thread copy the status word to pinned host memory
 * at that point, and dgr_early_status_wait() blocks until that copy -- not the rest of the forward -- has completed and
 * returns {num_rendered, 0, prefiltered violation, 0} (the overflow flag is the caller's own num_rendered > capacity).
 * Returns 1 when nothing was posted (P == 0). */
/* Asynchronous read-back of a device status word (the bindings' lazy mode): dgr_status_post enqueues a copy of
 * device_status[0..3] to pinned host memory behind an event on `stream` and returns a ticket (>= 0) or an error code;
 * dgr_status_poll returns 1 and fills host_status4 once the copy has landed (the ticket is then released), 0 when
 * `wait` == 0 and it has not yet.  dgr_stream_is_capturing: 1 while `stream` records a hipGraph (nothing can be read
 * back then). */
long dgr_status_post(void* stream, const int* device_status);
int dgr_status_poll(long ticket, int wait, int* host_status4);
/* The same read-back without the copy and the event: dgr_status_arm() returns a ticket and hands its slot to the NEXT
 * *_forward_presized call of this thread (whichever stream it runs on).  That forward's blend kernel (its first workgroup, before anything else) writes {num_rendered,
 * overflow, prefiltered violation, 0} straight into the slot's pinned host memory (mapped into the device's address space),
 * a tag last; dgr_status_poll on the ticket then reads host memory -- no HIP call unless it has to wait.  num_related_primitives
 * (full variant; completed by the forward blend, later than the rest) is NOT reported this way: read the device word when it
 * is needed.  A forward that enqueues no binning kernel (P == 0, an argument error) completes the word itself, all zero.  The
 * armed forward also reports its longest tile list, which feeds the "tile_schedule" policy below.  Not while `stream` records
 * a hipGraph (nothing can be read back then: do not arm).
 * A blocking poll (wait != 0) of an armed ticket cannot spin for ever: every millisecond it asks the forward's stream -- a HIP
 * error there ends the wait with DGR_ERR_HIP, and so does a stream that has finished all its work without the word having
 * arrived (a forward issued into a capturing stream, a faulted kernel) -- and it gives up after DGR_STATUS_TIMEOUT_MS
 * (environment, default 30 000; 0 = no limit: profiler replays and a collective's stragglers can hold a queue for longer).  The
 * stream is not asked while it records a hipGraph (a query would invalidate the capture).  The ticket is released on every such
 * return; after a timeout or a stream error its slot stays out of use until that stream has drained (the forward may still be
 * queued and would write into a slot that had been handed to another forward meanwhile). */
long dgr_status_arm(void);
int dgr_stream_is_capturing(void* stream);
int dgr_early_status_arm(void);
int dgr_early_status_wait(int* host_status4);

/* Fused sparse Adam step on one per-Gaussian tensor [rows, k] (the reference leaves the optimiser
 * to torch).  torch.optim.Adam's update (no weight decay, no amsgrad) with bias correction for the 1-based `step`;
 * rows with visible[row] <= 0 are skipped -- parameter and both moments untouched, as in 3DGS's sparse Adam -- and
 * `visible` == NULL updates every row.  `visible` is typically the forward's `radii`. */
int dgr_sparse_adam(void* stream, long rows, int k, float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                    const int* visible, float lr, float beta1, float beta2, float eps, int step);
/* The same with the 1-based step count read from device memory when the kernel runs, so that a step recorded into a
 * hipGraph keeps its bias correction right on every replay (the caller increments *step_device before, on the stream). */
int dgr_sparse_adam_capturable(void* stream, long rows, int k, float* param, const float* grad, float* exp_avg,
                               float* exp_avg_sq, const int* visible, float lr, float beta1, float beta2, float eps,
                               const int* step_device);

/* Densification bookkeeping for the rows one view saw (radii[row] > 0), as 3DGS's GaussianModel.add_densification_stats
 * and the training loop's max_radii2D update do with indexed torch ops (the reference's caller; the rasterizer only
 * hands back dL_dmeans2D for this purpose, diff_gaussian_rasterization/__init__.py L:164-176):
 *   grad_accum[row] += |dmeans2D[row, 0:2]|;  denom[row] += 1;  max_radii2D[row] = max(max_radii2D[row], radii[row]).
 * dmeans2D is [rows, 3]; each of the three outputs ([rows] floats) may be NULL to skip it. */
int dgr_densification_stats(void* stream, long rows, const float* dmeans2D, const int* radii, float* grad_accum, float* denom,
                            float* max_radii2D);

/* ---- the small pieces of a tracking iteration around the rasterizer, one launch each (the reference's
 * caller, CG-SLAM, does these with a dozen elementwise torch kernels each -- a 640x480 tracking step is launch-bound) ----
 * dgr_pose_forward: (quat = (r, x, y, z), not necessarily unit; trans) -> the three camera tensors the rasterizer takes,
 *   in its convention (cuda_rasterizer/auxiliary.h:58-77): viewmatrix = W2C^T, projmatrix = W2C^T * perspec_matrix with
 *   perspec_matrix = Proj^T, campos = -R^T t.  16 + 16 + 3 floats.
 * dgr_pose_backward: dL/dviewmatrix[16] (the rasterizer's pose gradient) -> dL/dquat[4], dL/dtrans[3], through the
 *   normalisation of quat.  projmatrix and campos are constants of the rasterizer call, as in the reference
 *   (diff_gaussian_rasterization/__init__.py L:164-176 returns a gradient for `viewmatrix` only). */
int dgr_pose_forward(void* stream, const float* quat, const float* trans, const float* perspec_matrix, float* viewmatrix,
                     float* projmatrix, float* campos);
int dgr_pose_backward(void* stream, const float* quat, const float* dL_dviewmatrix, float* dL_dquat, float* dL_dtrans);
/* loss = w_color * mean|color - color_obs| + w_depth * mean|depth - depth_obs| (two launches, fixed summation order);
 * `scratch` holds dgr_l1_loss_scratch_floats() floats.  The backward writes both gradient images in one launch:
 * w / n * sign(difference) * (*upstream), `upstream` == NULL meaning 1. */
int dgr_l1_loss_scratch_floats(void);
int dgr_l1_loss_forward(void* stream, long n_color, const float* color, const float* color_obs, long n_depth, const float* depth,
                        const float* depth_obs, float w_color, float w_depth, float* scratch, float* loss);
int dgr_l1_loss_backward(void* stream, long n_color, const float* color, const float* color_obs, long n_depth, const float* depth,
                         const float* depth_obs, float w_color, float w_depth, const float* upstream, float* dL_dcolor,
                         float* dL_ddepth);

/* Process-wide options (default 0 unless stated).
 *  "alpha_mode": how the blend kernels evaluate alpha = min(0.99, o exp(power)) and T / (1 - alpha).  0 (default) = the
 *     reference's expression in the reference's association (forward.cu:354-364, backward.cu:561-570) with an expf and a
 *     division whose bits the CPU restatement reproduces on any host (csrc/exact_math.h: an fp32-only polynomial expf,
 *     <= 0.9 ulp -- as faithful to nvcc's <= 2-ulp expf as any other; the correctly rounded quotient): the alpha image,
 *     n_contrib and the median depth are bit-identical to the restatement and the gradients agree with it to 1e-5 abs
 *     (BASELINE's loss scaling).
 *     1 = log2(e)-scaled conic, v_exp_f32, v_rcp_f32: each operation good to an ulp and the blend kernels faster, but
 *     the light backward's T_final = 1 - alpha image and its divisions by (1 - alpha) amplify last-bit differences: up to
 *     6e-5 abs on the pose gradient at BASELINE config 3.
 *     2 = as 0 with glibc's expf algorithm evaluated in the double pipe (rounds 5-7's default; slower), kept for A/B.
 *     Forward and backward of a view must run in the same mode.
 *  "fast_alpha": the older name of alpha_mode 0 / 1 (get: 1 iff alpha_mode == 1).
 *  "deterministic_grads": 1 = the light backward (one-view entry point, alpha_mode 0) forms its gradients without order-dependent
 *     float atomics: every quadrant wave of a tile keeps its own accumulators (added in wave order), every (tile, Gaussian) pair
 *     stores its finished row into an instance-major buffer (zero-filled first) and a Gaussian's rows are added in ascending tile
 *     order; the pose gradient's block partials are stored per block and added in block order.  Two runs give the same bits; the
 *     reference's own result, float atomics in arrival order (L/cuda_rasterizer/backward.cu:593-596, 666-680), lies within its
 *     run-to-run spread of it.  Costs about a quarter of the backward at config 3 (profiles/r8/deterministic.txt).  Needs
 *     dgr_light_backward_scratch_bytes_r() of scratch.  The full variant and the batched entry points refuse the option.
 *  "tight_cull": 1 = alpha-aware tile rectangles.  The reference gives a Gaussian every tile its
 *     3-sigma_max circle touches (cuda_rasterizer/forward.cu:229-237, auxiliary.h:46-56); with this option the rectangle
 *     is cut down to the box where alpha can reach 15/255.  Images and gradients are unchanged, but num_rendered, the
 *     tile lists and n_contrib are NOT the reference's any more -- hence opt-in.
 *  "tile_schedule" (default 2): whether the forward builds the blend kernels' tile schedule (classes of long lists first, so that
 *     a cluster's long lists start first and every XCD gets its share of them: 2x on the blend kernels of a clustered frame).
 *     1 = always, 0 = never (static map: every XCD a contiguous band of the image), 2 = by the frame: a forward whose status
 *     came back through dgr_status_arm also reports its longest tile list, and the next forward of that shape (device, P, width,
 *     height) skips the schedule kernel when that list was within twice the mean + 32 -- on an even frame the schedule is a
 *     launch and 11 us in front of the blend for nothing.  Forwards that report nothing keep it.  Results never depend on it.
 *  "lane_lists" (default 2): the lists the LIGHT blend kernels walk.  1 = one list per half of a quadrant wave (forward, tracking
 *     backward) and paired lists (mapping backward); 0 = one list per quadrant wave (rounds 1-7); 2 = by the FRAME, on the device:
 *     the binning kernels flag a frame of big splats (more than ten tiles per Gaussian on screen, profiles/r9/lists_sweep.txt), where
 *     nearly every entry lives in both halves of its quadrant and the finer lists cost 3 % for nothing, in the frame's state;
 *     forward and backward branch on that word.  Images never depend on it; gradients differ by summation order only.
 *     (DGR_FWD_HALVES = 0 / 1 in the environment sets the initial value.)
 *  "lds_count" (default 1): how the forward bins a frame's tile instances.  1 = the two-level segment binning
 *     (csrc/segment_binning.hip: pairs per 16-, 8- or 4-tile row segment, tile lists built and sorted in LDS; no global
 *     atomics, no cleared counters) whenever the frame's segment tables fit LDS (up to 8 192 four-tile segments, i.e.
 *     3840x2160 and a little beyond); 0 = returning global atomics on per-tile counters (csrc/binning.hip, round 2's path),
 *     which also serves larger frames.  Results are identical bit for bit.  dgr_binning_bytes() includes the segment
 *     binning's forward-only tables.
 *  "blend_wgs_per_cu" (default 0 = no cap; 3..7): at most this many blend workgroups per CU (they are padded with unused
 *     LDS), which leaves wave slots, registers and LDS for kernels of other streams -- collectives, other views' bandwidth-
 *     bound kernels -- that otherwise enter a CU only as blend workgroups drain.  Costs the blend kernels 3-7 % at 7.
 *  "profile_every": n >= 1 = dgr_profile_* brackets every n-th launch of the selected stage only (default 1).
 *  "batch_streams" (default 2): streams the batched entry points spread the per-view stages of a batch over (1..8).
 *  "batch_order" (default 0): 0 = view v's binning + blend chain on stream v mod batch_streams; 1 = all binning stages on a
 *     helper stream and all blend stages on the caller's, joined per view by events (measured slower: csrc/api.hip). */
int dgr_set_option(const char* name, int value);
int dgr_get_option(const char* name);

/* Per-THREAD values of the options that change what a call computes -- "alpha_mode" (and its older name "fast_alpha"),
 * "tight_cull", "deterministic_grads" -- overriding the process-wide ones above for the calling thread's next calls (value < 0:
 * inherit again).  A tracker and a mapper thread of one process hold different settings this way, and nothing a thread sets
 * reaches launches that are already queued: every entry point reads its options once, when it is called.  (The reference has
 * no options; its one compile-time choice is the variant.)  dgr_get_thread_option = the value the calling thread's next call
 * uses.  dgr_thread_options_effective() packs the three (each field value + 1: bits 0-3 alpha_mode, 4-7 tight_cull, 8-11
 * deterministic_grads) and dgr_thread_options_swap(word) installs such a word as the thread's overrides (field 0 = inherit;
 * word < 0: only read) and returns the previous one -- what an autograd binding uses to run a backward, on whatever thread the
 * engine picks, under its forward's options. */
int dgr_set_thread_option(const char* name, int value);
int dgr_get_thread_option(const char* name);
int dgr_thread_options_effective(void);
int dgr_thread_options_swap(int word);

/* ---- work shared by the views of a batch ----
 * The 3D covariance depends on scale and rotation only.  dgr_cov3d_forward evaluates computeCov3D (cr/forward.cu:118-152)
 * once -- bit-identical to what the forward would compute per view -- for use as `cov3D_precomp` of every view of the batch;
 * dgr_cov3d_backward is its backward (L/cr/backward.cu:280-343), linear in dL_dcov3D: sum the views' dL_dcov3D first, convert
 * once.  scales [P,3], rotations [P,4] (r,x,y,z; not normalised), cov3D / dL_dcov3D [P,6], dL_dscales [P,3], dL_drotations [P,4]. */
int dgr_cov3d_forward(void* stream, int P, const float* scales, const float* rotations, float scale_modifier, float* cov3D);
int dgr_cov3d_backward(void* stream, int P, const float* scales, const float* rotations, float scale_modifier,
                       const float* dL_dcov3D, float* dL_dscales, float* dL_drotations);

/* ---- batched multi-view entry points (BASELINE configs 4 and 5: several cameras over ONE set of Gaussians) ----
 * The reference renders a keyframe batch as V independent Rasterizer::forward / backward calls (L/cr/rasterizer.h:40-104) and
 * lets autograd add the V dense gradient sets.  Here one call takes the V cameras; per view it does exactly what
 * dgr_light_forward_presized / dgr_light_backward do (every view's outputs and state buffers are bit-identical to a one-view
 * call, and usable by one), but the work that does not depend on the camera is shared:
 *  - forward: ONE per-Gaussian launch for all views -- position, opacity, 3D covariance formed once, the 192-byte SH row
 *    fetched once and evaluated for every camera that sees the Gaussian;
 *  - backward: ONE per-Gaussian launch that sums the views' gradients of the shared Gaussians in registers and writes each
 *    dense row (248 bytes per Gaussian at SH degree 3) once instead of V times plus V - 1 accumulation passes; the covariance
 *    backward (linear in dL_dcov3D) runs once on the sum.  Given the same per-view accumulator rows, dL_dopacity / dL_dmean3D /
 *    dL_dsh / dL_dcov3D are the one-view outputs accumulated in view order, operation for operation; dL_dscale / dL_drot agree
 *    to rounding (converted once, not V times).  (Two runs of the blend backward -- batched or not -- leave accumulator rows
 *    that differ by the arrival order of their float atomics, so two runs agree to ~1e-6 of a tensor's scale, not bitwise);
 *  - the per-view stages in between (binning and blend) are issued on internal streams forked from and joined to `stream`
 *    with events (option "batch_streams", 1..8, default 2: one view's binning runs under another view's blend);
 *  - no host synchronisation (hipGraph-capturable after one warm-up call, which creates the internal streams).
 * All views share the image size, tan_fovx / tan_fovy, background and the Gaussians; `views` is a HOST array of n_views
 * (1 .. DGR_MAX_BATCH_VIEWS) structs of DEVICE pointers, read during the call only.  Light variant. */
#define DGR_MAX_BATCH_VIEWS 8
typedef struct dgr_light_view {        /* the per-camera arguments of dgr_light_forward_presized, same meaning */
    char* geometry_buffer;
    char* binning_buffer;
    int binning_capacity;
    char* image_buffer;
    int* status;                        /* device int[4] or NULL */
    const float* viewmatrix;
    const float* projmatrix;
    const float* cam_pos;
    float* out_color;
    float* out_depth;
    float* out_median_depth;
    float* out_alpha;
    const float* gt_depth;
    float* out_depth_var;
    float* gau_uncertainty;
    int* gau_related_pixels;
    int* radii;                         /* may be NULL */
} dgr_light_view;
int dgr_light_forward_batch(void* stream, int n_views, const dgr_light_view* views, int P, int D, int M,
                            const float* background, int width, int height, const float* means3D, const float* shs,
                            const float* colors_precomp, const float* opacities, const float* scales, float scale_modifier,
                            const float* rotations, const float* cov3D_precomp, float tan_fovx, float tan_fovy, int prefiltered);

typedef struct dgr_light_view_grad {   /* the per-camera arguments of dgr_light_backward, same meaning */
    char* geometry_buffer;
    char* binning_buffer;
    char* image_buffer;
    const float* viewmatrix;
    const float* projmatrix;
    const float* cam_pos;
    const float* perspec_matrix;
    const float* alphas;
    const float* gt_depth;
    const int* radii;                   /* may be NULL (internal copy is used) */
    const float* dL_dpix;
    const float* dL_dpix_depth;
    const float* dL_dpix_median_depth;
    const float* dL_dpix_depth_var;
    float* dL_dmean2D;                  /* [P,3] of THIS view (densification statistics are per view); may be NULL */
    float* dL_dview;                    /* [16] of this view */
    char* scratch;                      /* dgr_light_backward_scratch_bytes(), one per view (deterministic_grads:
                                           dgr_light_backward_scratch_bytes_r(P, width, height, num_rendered)) */
    size_t scratch_bytes;
    int num_rendered;                   /* this view's R as passed to dgr_light_backward (>= its num_rendered); read only with
                                           deterministic_grads, where it sizes the view's instance-major row buffer */
} dgr_light_view_grad;
/* dL_dopacity [P], dL_dcolor [P,3] (may be NULL), dL_dmean3D [P,3], dL_dcov3D [P,6] (may be NULL), dL_dsh [P,M,3] (NULL when
 * M == 0), dL_dscale [P,3], dL_drot [P,4]: the SUM over the views.  Every one of them may be NULL (tracking: map_off = 1). */
int dgr_light_backward_batch(void* stream, int n_views, const dgr_light_view_grad* views, int P, int D, int M,
                             const float* background, int width, int height, const float* means3D, const float* shs,
                             const float* colors_precomp, const float* scales, float scale_modifier, const float* rotations,
                             const float* cov3D_precomp, float tan_fovx, float tan_fovy, float* dL_dopacity, float* dL_dcolor,
                             float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot,
                             int track_off, int map_off);

/* Debug: while `device_words` (8 x uint64 per bin_tiles workgroup, caller-owned device memory) is non-NULL, every bin_tiles
 * workgroup stores phase time stamps (100 MHz wall clock) and its segment's sizes there: profiles/r9/bin_tiles_trace.py. */
int dgr_debug_bin_tiles_trace(unsigned long long* device_words);

/* Self-test of the wave64 multi-value butterfly reductions the backward blend relies on (csrc/wave_reduce.h: within-row DPP
 * stages first, then v_permlane16/32_swap as inline asm).  `in` holds 16 values per lane as in[c * 64 + lane]; out16 /
 * out12 / out4 [lane] receive what each lane holds after the 16- / 12- / 4-value network (the 12- and 4-value ones read the
 * first 12 / 4 values), comp16 / comp12 / comp4 [lane] the index of the value that lane's total belongs to.  64 entries each. */
int dgr_debug_wave_reduce(void* stream, const float* in, float* out16, float* out12, float* out4, int* comp16, int* comp12,
                          int* comp4);
/* Self-test of the reductions per HALF of a wave (csrc/wave_reduce.h), which the blend backward uses where a loop step serves one
 * list entry on lanes 0-31 and another on lanes 32-63.  `in` holds 12 values per lane as in[c * 64 + lane].  r0 / r1 [lane]: what
 * the lane holds after the twelve-value network stopped before its cross-half stage, slot0 / slot1 [lane] the index of the value
 * that total belongs to (-1: none); h3 [lane]: after the three-value network of the tracking backward (the first three values),
 * comp3 [lane] the value index (-1: none).  64 entries each. */
int dgr_debug_half_reduce(void* stream, const float* in, float* r0, float* r1, float* h3, int* slot0, int* slot1, int* comp3);
/* ... and of the SIXTEEN-value network stopped before its cross-half stage (the paired step of the full variant's backward): `in`
 * holds 16 values per lane as in[c * 64 + lane]; r0 / r1 [lane] what the lane holds, slot0 / slot1 [lane] the value index. */
int dgr_debug_half_reduce16(void* stream, const float* in, float* r0, float* r1, int* slot0, int* slot1);
/* Self-test of the per-wave list builders of the blend kernels (csrc/render_common.h: build_paired_lists, build_half_lists) on one
 * batch of 128 staged slots.  codes[slot] (128 device bytes): bit 2 w + h set <=> half h (lanes 32 h ..) of quadrant wave w takes
 * the slot.  paired / halves (4 x 280 device words, one block per wave): {steps or length, split mask 0 lo, hi, split mask 1 lo,
 * hi, list 2 w [136], list 2 w + 1 [136]}; list entries are record offsets (32 x slot), 32 x 128 = the sentinel. */
int dgr_debug_lane_lists(void* stream, const unsigned char* codes, unsigned* paired, unsigned* halves);
/* Self-test of csrc/exact_math.h, the arithmetic behind the default alpha path: out_exp[i] = the current alpha mode's expf of
 * x[i] (mode 0: exp_p32, mode 2: exp_glibc; x <= 0, clamped at -104) and out_div[i] = div_ref(a[i], b[i]) -- correctly rounded
 * a / b for normal operands.  n device floats each.  tests/test_hip_exact_math.py compares both with the CPU restatement's
 * functions bit for bit. */
int dgr_debug_exact_math(void* stream, int n, const float* x, const float* a, const float* b, float* out_exp, float* out_div);

/* ---- per-stage timing (bench.py's roofline object) ----
 * dgr_profile_select("") disables timing (default), "all" brackets every stage, a stage name brackets that
 * stage only with two HIP events recorded on the launching stream.  Stage names: dgr_profile_stage_name(i),
 * i < dgr_profile_stage_count().  dgr_profile_read waits for the recorded events, returns their summed
 * duration and the number of launches since the previous read, and clears them. */
int dgr_profile_select(const char* stage);
int dgr_profile_stage_count(void);
const char* dgr_profile_stage_name(int i);
int dgr_profile_read(const char* stage, double* total_ms, int* launches);

#ifdef __cplusplus
}
#endif
#endif /* DGR_HIP_H */
