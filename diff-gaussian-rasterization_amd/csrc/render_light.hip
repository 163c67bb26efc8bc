// render_light.hip -- 16x16-tile alpha-blending kernels of the light variant for gfx950.
//
// Replaces renderCUDA forward (L/cuda_rasterizer/forward.cu:261-412) and renderCUDA backward
// (L/cuda_rasterizer/backward.cu:419-699).
//
// Shape of the computation on CDNA4 (wave64, LDS 160 KB/CU, ONE scalar unit per CU):
//  * one 256-thread workgroup per tile, wave w owns the 8x8-pixel quadrant (w&1, w>>1);
//  * the sorted tile list is staged 256 instances at a time into LDS: a 32-byte traversal record
//    {x, y, a2, b2 | c2, opacity, slot, -} plus {r, g, b, depth} and the Gaussian id, one 48-byte gather (one 64-byte line)
//    per thread;
//  * while staging, every thread bounds the region where its Gaussian can reach alpha >= 15/255 (the
//    ellipse q(d) <= 2 ln(255 o / 15), boxed with a safety margin) and tests it against the eight
//    HALVES of the four quadrants (8x4 pixels = lanes 0-31 / 32-63 of the consumer wave).  Wave ballots +
//    mbcnt turn those tests into eight COMPACTED lists of 16-bit record offsets, one per half-wave, in
//    tile-list order (the backward: four lists, one per wave, from the forward's contribution tags);
//  * each wave walks its two lists side by side, two entries per iteration: one 4-byte LDS read per lane
//    yields two offsets, ds_read_b128 with one address per half-wave fetch the records (the same four
//    LDS cycles as a broadcast), then pure VALU per pixel.  A splat reaches 16 of a quadrant's 64
//    pixels on average and more than half of the pairs stay inside one half, so the halves need 0.73
//    of the steps a quadrant-wide list takes (render_fwd 112 -> 103 us).  The loop has no scalar bit
//    scans and "pixel finished" is a per-lane threshold register rather than a lane mask: the first
//    version of this loop spent ~20 SALU instructions per Gaussian and was bound by the CU's single
//    scalar unit (SQ_INSTS_SALU ~ 0.87 per CU cycle), not by the four SIMDs.
//    A skipped (pixel, Gaussian) pair is one the per-pixel test `power > 0 || alpha < 15/255` would have
//    rejected, so outputs are unchanged; `contributor` (hence n_contrib) is the position in the tile list,
//    which skipping does not alter.
//  * block -> tile through the tile schedule (ImageView::tile_sched, binning.hip): longest list first, consecutive workgroups
//    going round the XCDs.  (Rounds 1-5 handed every XCD a contiguous band of tiles for L2 reuse between neighbours: no
//    measurable gain, and a factor of two lost on a frame whose Gaussians cluster -- DESIGN.md s4.8.)
//
// alpha: template parameter AM (render_common.h).  ALPHA_REF, the default, evaluates the reference's expression with the
// CPU restatement's bits (exact_math.h: an fp32 polynomial expf; ALPHA_GLIBC: glibc's algorithm in the double pipe): the
// alpha image, n_contrib and the median depth then equal the restatement's bit for bit, which is what the light backward's
// T_final = 1 - alpha needs.  ALPHA_FAST (an option) is
// o * exp2(p2) on a conic pre-scaled by log2(e) -- one v_exp_f32.  Forward and backward of one mode use the identical
// expression, so they agree on every decision.
#include "render_common.h"

namespace dgr {
namespace {

// ================================================================================ forward
constexpr int FWD_UNROLL = 2;  // list entries per loop iteration (4 was measured: no faster, more registers)

// HALVES: eight lists, one per half of a quadrant (render_common.h: build_half_lists); the ids stay out of LDS to keep 8 workgroups per CU
template <bool HALVES>
struct StagedFwd {
    typedef StagedT<DGR_TILE_PIX, unsigned short, HALVES ? 8 : 4, !HALVES> staged_t;
    staged_t f;
    float unc[DGR_TILE_PIX];   // per staged instance: sum of (d - gt)^2 alpha T over its median pixels (forward.cu:386)
    uint32_t cnt[DGR_TILE_PIX];
    uint32_t hit[DGR_TILE_PIX];  // byte w of word j != 0 <=> some pixel of quadrant wave w blended staged instance j
    uint64_t exptab[32];         // ALPHA_GLIBC: exact_math.h
};

// Per-slot results of the batch staged at list position `pos0`: the median statistics go to the Gaussian, the
// contribution tag into the entry's tag byte (render_common.h).

template <class SF>
__device__ __forceinline__ void flush_slot(const SF& sf, const RenderFwdLightArgs& a, uint8_t* tag8, uint32_t pos0, int tid, bool staged) {
    if (!staged) return;
    uint32_t t8;  // bit 2 w <- upper half of wave w, bit 2 w + 1 <- its lower half
    if constexpr (!SF::staged_t::HAS_ID) {  // (the half-wave body) lower halves: the bytes of the record's third word
        t8 = tag_byte(sf.hit[tid], __float_as_uint(sf.f.rec[2 * tid + 1].z));
    } else {                                // (the quadrant body) a quadrant's tag stands for both of its halves
        t8 = spread4(pack4(sf.hit[tid])) * 3u;
    }
    tag8[pos0 + tid] = (uint8_t)t8;         // every staged entry, blended or not: the byte underneath is the binning's
    if (sf.cnt[tid] != 0u) {                // (which implies a tag)
        uint32_t gid;
        if constexpr (SF::staged_t::HAS_ID) gid = sf.f.id[tid];
        else gid = a.point_list[pos0 + tid];
        atomicAdd(&a.gau_uncertainty[gid], sf.unc[tid]);
        atomicAdd(&a.gau_related_pixels[gid], (int)sf.cnt[tid]);
    }
}

template <int AM, bool HALVES>
__device__ __forceinline__ void render_fwd_light_body(const RenderFwdLightArgs& a, StagedFwd<HALVES>& sf, const uint4 slot, const bool overflowed) {
    typename StagedFwd<HALVES>::staged_t& s = sf.f;
    const int tile = (int)slot.x;
    const int tx = tile % a.grid_x, ty = tile / a.grid_x;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int px = tx * DGR_BLOCK_X + (wave & 1) * 8 + (lane & 7);
    const int py = ty * DGR_BLOCK_Y + (wave >> 1) * 8 + (lane >> 3);
    const bool inside = px < a.W && py < a.H;
    const uint32_t pix_id = (uint32_t)a.W * (uint32_t)py + (uint32_t)px;  // (W H <= 2^30: api.hip)
    const f2 pxy = {(float)px, (float)py};
    const float tile_x0 = (float)(tx * DGR_BLOCK_X), tile_y0 = (float)(ty * DGR_BLOCK_Y);
    const int my_list = HALVES ? 2 * wave + (lane >> 5) : wave;
    // where this lane marks "blended": byte `wave` of hit[j] -- HALVES: lanes 32-63 in byte `wave` of the record's spare word
    uint8_t* const tag8 = half_tags(a.point_list, a.sched_flag);
    unsigned char* const mark_base = (HALVES && lane >= 32) ? reinterpret_cast<unsigned char*>(&s.rec[1].z) + wave
                                                             : reinterpret_cast<unsigned char*>(sf.hit) + wave;
    const int mark_stride = (HALVES && lane >= 32) ? 32 : 4;

    const uint2 range = make_uint2(slot.y, slot.z);
    const int total = (int)(range.y - range.x);

    float T = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f, weight = 0.f, Dd = 0.f, D_median = 0.f;
    uint32_t last_contributor = 0;
    // "done" as a per-lane upper bound on p2: live pixels accept p2 <= 0 (the reference's `power > 0` test),
    // finished pixels compare against -inf and accept nothing
    float ub = inside ? 0.f : -__builtin_inff();
    const float gt_px = inside ? a.gt_depth[pix_id] : 0.f;
    if (tid == 0) write_sentinel(s);
    if (AlphaPath<AM>::TABLE) exp_ref_table_fill(sf.exptab, tid);  // (visible after the first batch's barriers)
    bool have_flush = false;
    int last_base = 0;

    for (int base = 0; base < total; base += DGR_TILE_PIX) {
        // whole tile finished?  (L/cuda_rasterizer/forward.cu:329-332)
        if (__syncthreads_and(ub < 0.f)) break;
        last_base = base;
        // median statistics of the previous batch: slot tid is flushed by the thread that restages it
        if (have_flush) flush_slot(sf, a, tag8, range.x + base - DGR_TILE_PIX, tid, true);  // (an earlier batch is always full)
        sf.unc[tid] = 0.f;
        sf.cnt[tid] = 0u;
        sf.hit[tid] = 0u;
        have_flush = true;
        const int cnt = min(DGR_TILE_PIX, total - base);
        unsigned code = 0;
        if (tid < cnt) code = stage_one<AM, HALVES>(s, tid, a.point_list[range.x + base + tid], a.rec, tile_x0, tile_y0);
        const int n = HALVES ? build_half_lists(s, code, tid, wave, lane) : build_lists(s, code, tid, wave, lane);

        for (int k = 0; k < n; k += FWD_UNROLL) {
            float4 q0[FWD_UNROLL], q1[FWD_UNROLL];
            load2(s, my_list, k, q0, q1);
#pragma unroll
            for (int u = 0; u < FWD_UNROLL; u++) {
                f2 dxy;
                const float p2 = pair_p2<AM>(q0[u], q1[u], pxy, dxy);
                if ((p2 <= ub) & (p2 >= q1[u].w)) {  // cheap log-domain pre-test: the exponential stays off the common path
                  const float alpha = fminf(0.99f, alpha_raw<AM>(q1[u].y, p2, sf.exptab));
                  if (alpha >= ALPHA_MIN) {
#pragma clang fp contract(off)  // T (1 - alpha) and the alpha image's sum round as the reference's do (forward.cu:366-379)
                    const float test_T = T * (1.0f - alpha);
                    if (test_T < 0.0001f) {
                        ub = -__builtin_inff();  // done; this Gaussian is not blended (forward.cu:368-373)
                    } else {
                        const int j = HALVES ? (__float_as_int(q1[u].w) & 0xFF) : __float_as_int(q1[u].z);
                        const float4 cd = s.rgbd[j];
                        mark_base[j * mark_stride] = 1;  // contribution tag
                        const float w = alpha * T;
                        C0 = __builtin_fmaf(cd.x, w, C0); C1 = __builtin_fmaf(cd.y, w, C1); C2 = __builtin_fmaf(cd.z, w, C2);
                        weight = weight + w;
                        Dd = __builtin_fmaf(cd.w, w, Dd);
                        if (T > 0.5f && test_T < 0.5f) {  // forward.cu:381-388
                            D_median = cd.w;
                            const float e = cd.w - gt_px;
                            atomicAdd(&sf.unc[j], e * e * w);
                            atomicAdd(&sf.cnt[j], 1u);
                        }
                        T = test_T;
                        last_contributor = (uint32_t)(base + j + 1);
                    }
                  }
                }
            }
            if (!wave_any(ub >= 0.f)) break;
        }
    }
    __syncthreads();
    if (have_flush) flush_slot(sf, a, tag8, range.x + last_base, tid, tid < total - last_base);
    // the tail of a list whose tile finished early was never staged: nobody blended it (render_common.h: the tag bytes' invariant)
    for (int p = (have_flush ? last_base + DGR_TILE_PIX : 0) + tid; p < total; p += DGR_TILE_PIX) tag8[range.x + p] = 0;

    // A forward whose binning buffer was too small has rendered EMPTY tile lists (bin_tiles left every range {0, 0}): in lazy mode
    // the host learns of it a call or two later, so the images must not look like a frame -- they are NaN, every value
    // (strict mode retries inside the call and overwrites them).  The flag came with the schedule word (blend_slot): a uniform branch.
    if (overflowed) {
        C0 = C1 = C2 = weight = Dd = D_median = __builtin_nanf("");
    }
    if (inside) {
        const uint32_t N = (uint32_t)a.W * (uint32_t)a.H;
        a.n_contrib[pix_id] = last_contributor;
        a.out_color[pix_id] = C0 + T * a.bg[0];
        a.out_color[N + pix_id] = C1 + T * a.bg[1];
        a.out_color[2 * N + pix_id] = C2 + T * a.bg[2];
        a.out_alpha[pix_id] = weight;  // forward.cu:407
        a.out_depth[pix_id] = Dd;
        a.out_median[pix_id] = D_median;
        a.out_depth_var[pix_id] = 0.0f;  // forward.cu:317,410
    }
}

// One kernel, both lane mappings: which one a frame takes is the frame's own flag (ImageView::cursor[3] bit 2, written by its binning
// kernel) -- a uniform branch on a word the workgroup reads anyway, so that the choice can be made on the DEVICE for the frame at
// hand (no host policy, nothing to carry from forward to backward: the backward kernels branch on the same word).  The two bodies
// share the workgroup's LDS (a union: 20.0 KB, 8 workgroups per CU) and the register budget of the launch bounds.
template <int AM>
__global__ void __launch_bounds__(256, 8) render_fwd_light_kernel(RenderFwdLightArgs a) {
    __shared__ union { StagedFwd<true> h; StagedFwd<false> q; } sf;
    if (a.rep.host && blockIdx.x == 0 && threadIdx.x == 0) report_status(a.rep, a.status);
    bool overflowed, quadrant_lists;
    const uint4 slot = blend_slot(a.sched, a.ranges, a.sched_flag, a.grid_x * a.grid_y, &overflowed, &quadrant_lists);  // {tile, list start, list end}
    if (quadrant_lists) render_fwd_light_body<AM, false>(a, sf.q, slot, overflowed);
    else render_fwd_light_body<AM, true>(a, sf.h, slot, overflowed);
}

// ================================================================================ backward
// Per-Gaussian gradient sums.  Only ~6 of a wave's 64 pixels are hit by any one Gaussian and LDS float
// atomics retire barely one lane per cycle on gfx950 (measured: 55 % of wave cycles in SQ_WAIT_INST_LDS
// with per-lane ds_add_f32), so the sums are formed in registers: every listed Gaussian is visited
// wave-uniformly (as in the forward), the valid lanes compute their 14 contributions, and ONE butterfly
// (wave_reduce.h, 33 instructions) reduces all 14 across the wave at once, leaving each total in its own
// lane quad.  A single ds_add_f32 with 14 active lanes on 14 distinct banks then merges the four
// quadrant waves in LDS accumulators acc[14][257], which are flushed once per batch with line-coalesced
// global atomics (16 consecutive lanes = one Gaussian's 64-byte accumulator row).  Global float atomics
// drop from 14 per valid (pixel, Gaussian) pair to <= 14 per (tile, Gaussian).
// Tracking mode (map_off) needs only the three sums the pose gradient is built from: a 4-value butterfly.
// (Twelve values since round 3 -- raw moments, the median term as the twelfth -- in accumulator rows 0..10 and 13.)
// Round 8: a Gaussian reaches 16 of a quadrant's 64 pixels on average and more than half of the (quadrant, Gaussian) entries
// live in ONE half of the quadrant.  The forward tags what it blended per half (render_common.h: half_tags); from those tags
// the tracking kernel walks one list per half-wave like the forward, and the mapping kernel pairs neighbouring entries of its
// list that live in different halves into one loop step (build_paired_lists) -- see the HALVES note above the kernel.
constexpr int NACC_LIGHT = 14;

// list positions staged per batch (256: 5 workgroups per CU, 267 us; 128: 247 us).  The deterministic kernel (DET, below) keeps one
// accumulator plane per quadrant wave -- four times the accumulators -- and stages 64 positions per batch to stay at 8 workgroups
// per CU.
template <bool DET, bool HALVES = false>
struct StagedBwd {
    static constexpr int NB = DET ? 64 : 128;
    static constexpr int LD = NB + 1;          // accumulator row length
    static constexpr int PLANE = NACC_LIGHT * LD;
    typedef StagedT<NB, uint32_t, HALVES ? 8 : 4> staged_t;
    staged_t f;
    float acc[(DET ? 4 : 1) * PLANE];
    uint32_t inst[DET ? NB : 1];               // DET: the staged entries' rows in the instance-major gradient buffer (~0u: none)
    int max_last;
    uint64_t exptab[32];  // ALPHA_GLIBC: exact_math.h
};

// 8 waves per SIMD (63 VGPRs, no scratch): measured 258 us at the compiler's own choice of 7, 247 us at 8
//
// DET (dgr_set_option("deterministic_grads", 1)): gradients that are the same bits run after run.  Two sums of the default kernel
// depend on the order in which hardware atomics arrive: the four quadrant waves' totals of a (tile, Gaussian) pair meet in LDS
// float atomics, and the tiles' totals of a Gaussian meet in global float atomics (as in the reference, backward.cu:593-596,
// 666-680).  Here (a) every wave stores its total into its OWN accumulator plane -- a wave visits a list entry once per batch, so
// a plain store suffices -- and the planes are added in wave order; (b) the finished row of a (tile, Gaussian) pair is STORED to
// the pair's own 64-byte row of an instance-major buffer (row = the Gaussian's first instance, det_offsets_kernel, + the tile's
// position in its rectangle; the buffer is zero-filled first: pairs nobody blended stay zero), and det_gather_kernel adds a
// Gaussian's rows -- contiguous -- in ascending order into the accumulator row preprocess_bwd reads.
//
// LEAN: the caller passed no gradient image for the median depth and none for the depth variance (NULL pointers: the loss did
// not use those outputs -- the usual case in CG-SLAM, whose depth_var output is identically zero, forward.cu:317,410).  The
// kernel then drops what only they feed: the variance term of X and of the depth gradient, and the once-per-pixel median test --
// five full-rate and three 4.2-cycle instructions per list entry.  Bit-identical to the full kernel fed all-zero images
// (0 * e^2 adds an exact zero).
//
// HALVES (the tracking backward, map_off): one list per HALF-wave and a loop step that serves the upper half's next entry on
// lanes 0-31 and the lower half's on lanes 32-63, as in the forward.  The lists come from the forward's tags per half
// (render_common.h: half_tags); each half reduces its three sums over its own 32 lanes
// (half_reduce3) and delivers them to its own entry's accumulator column: six lane-atomics per step.  The MAPPING backward
// delivers twelve values per entry and is bound by the LDS array's float atomics in this form.
//
// With DO_MAP, HALVES means PAIRED lists (render_common.h: build_paired_lists): neighbouring entries of a wave's list that live
// in different halves of the quadrant share a loop step (0.87 steps per entry); such a step reduces its twelve sums per half and
// each half delivers to its own entry's column -- every entry is still delivered once.
template <int AM, bool DO_MAP, bool DO_POSE, bool DET, bool LEAN, bool HALVES>
__device__ __forceinline__ void render_bwd_light_body(const RenderBwdLightArgs& a, StagedBwd<DET, HALVES>& sb, const uint4 slot) {
    static_assert(!HALVES || !DET, "half-wave / paired lists: not with the deterministic kernel's planes (LDS)");
    constexpr bool PAIRED = HALVES && DO_MAP;
    typedef StagedBwd<DET, HALVES> SB;
    constexpr int BWD_NB = SB::NB, BWD_LD = SB::LD;
    typename SB::staged_t& s = sb.f;
    const int tile = (int)slot.x;
    const int tx = tile % a.grid_x, ty = tile / a.grid_x;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int px = tx * DGR_BLOCK_X + (wave & 1) * 8 + (lane & 7);
    const int py = ty * DGR_BLOCK_Y + (wave >> 1) * 8 + (lane >> 3);
    const bool inside = px < a.W && py < a.H;
    const uint32_t pix_id = (uint32_t)a.W * (uint32_t)py + (uint32_t)px;  // (W H <= 2^30: api.hip)
    const uint32_t N = (uint32_t)a.W * (uint32_t)a.H;
    const float pxf = (float)px, pyf = (float)py;
    const f2 pxy = {pxf, pyf};

    const uint2 range = make_uint2(slot.y, slot.z);
    const int last_contributor = inside ? (int)a.n_contrib[pix_id] : 0;

    if (tid == 0) {
        sb.max_last = 0;
        write_sentinel<true>(s);
    }
    if (AlphaPath<AM>::TABLE) exp_ref_table_fill(sb.exptab, tid);
    __syncthreads();
    {
        int v = last_contributor;  // wave max, then one LDS atomic per wave
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off, 64));
        if (lane == 0) atomicMax(&sb.max_last, v);
    }
    __syncthreads();
    const int total = min((int)(range.y - range.x), sb.max_last);  // nothing past the last contributor matters
    if (total <= 0) return;

    const float T_final = inside ? (1.0f - a.alphas[pix_id]) : 0.f;
    float T = T_final;
    float dpix0 = 0.f, dpix1 = 0.f, dpix2 = 0.f, dpix_depth = 0.f, dpix_median = 0.f, dpix_var = 0.f, gt_px = 0.f;
    if (inside) {
        // (single-use images, 28 bytes per pixel: nontemporal, so that they do not push the accumulator rows and render records out of
        //  the caches -- preprocess_bwd behind this kernel 45 -> 43.5 us, profiles/r6/ab_nontemporal.txt)
        dpix0 = __builtin_nontemporal_load(a.dL_dpix + pix_id);
        dpix1 = __builtin_nontemporal_load(a.dL_dpix + N + pix_id);
        dpix2 = __builtin_nontemporal_load(a.dL_dpix + 2 * N + pix_id);
        dpix_depth = __builtin_nontemporal_load(a.dL_dpix_depth + pix_id);
        if (!LEAN) {  // (either image may be missing on its own: it then reads as zero)
            if (a.dL_dpix_median) dpix_median = __builtin_nontemporal_load(a.dL_dpix_median + pix_id);
            if (a.dL_dpix_var) dpix_var = __builtin_nontemporal_load(a.dL_dpix_var + pix_id);
            gt_px = __builtin_nontemporal_load(a.gt_depth + pix_id);
        }
    }
    // per-pixel constants of the loop: -T_final <bg, dL/dpixel> (the background term of dL/dalpha is this times
    // 1/(1 - alpha)) and 2 dL/dvar
    const float bg_term = -T_final * (a.bg[0] * dpix0 + a.bg[1] * dpix1 + a.bg[2] * dpix2);
    const float ddelx_dx = 0.5f * a.W, ddely_dy = 0.5f * a.H;
    // The reference keeps five "accumulated behind me" recurrences (3 colours, depth, variance: backward.cu:580-608)
    // only to form dL/dalpha = sum_c (c_j - accum_rec_c) dL/dpixel_c.  They are linear, so one scalar suffices:
    //   X_j = <features_j, dL/dpixel>,   S <- alpha_last X_last + (1 - alpha_last) S,   dL/dalpha = X_j - S.
    float S = 0.f;
    float mid_thr = 0.5f;
    // which accumulator component this lane's quad delivers after the butterfly (-1: none)
    int my_comp;
    if (DO_MAP) {
        // butterfly slots 0..9 = components 0..9; with the pose gradient slot 10 = component 13 (pose depth) and slot 11 =
        // component 10 (median), without it slot 10 = component 10
        const int c = (lane >= 48) ? 12 : wave_reduce12d_comp(lane);  // (rows 2 and 3 hold the same four totals: row 2 delivers)
        my_comp = ((lane & 3) != 0 || c > 11) ? -1 : (c == 10 ? (DO_POSE ? 13 : 10) : c == 11 ? (DO_POSE ? 10 : -1) : c);
    } else if (HALVES) {  // (tracking)
        const int c = half_reduce3_comp(lane);  // {4: gmx, 5: gmy, 13: pose depth} of the lane's half
        my_comp = c == 0 ? 4 : c == 1 ? 5 : c == 2 ? 13 : -1;
    } else {
        const int c = wave_reduce4_comp(lane);  // {4: gmx, 5: gmy, 13: pose depth}
        my_comp = ((lane & 15) == 0 && c < 3) ? (c == 0 ? 4 : c == 1 ? 5 : 13) : -1;
    }
    const int my_list = HALVES ? 2 * wave + (lane >> 5) : wave;
    const uint8_t* const tag8 = half_tags(a.point_list, a.sched_flag);

    // this lane's accumulator row (column = slot); DET: in its wave's own plane
    float* const my_acc = sb.acc + (DET ? wave * SB::PLANE : 0) + (my_comp >= 0 ? my_comp : 0) * BWD_LD;

    // back-to-front: batches cover list positions [lo, hi) with hi walking down from `total`
    for (int hi = total; hi > 0; hi -= BWD_NB) {
        const int lo = max(0, hi - BWD_NB);
        const int cnt = hi - lo;
        __syncthreads();  // previous batch fully flushed / consumed
        unsigned code = 0;
        if (tid < cnt) code = stage_tagged<AM, HALVES ? TAGS_BYTES_HALVES : TAGS_BYTES_QUADRANT>(s, tid, a.point_list[range.x + lo + tid], a.rec, tag8 + (range.x + lo + tid));
        if (!DET) {  // (DET: a plane's column is written by its wave iff the entry's tag names the wave -- nothing to clear)
#pragma unroll
            for (int k = 0; k < NACC_LIGHT; k++)
                if (BWD_NB == DGR_TILE_PIX || tid < BWD_NB) sb.acc[k * BWD_LD + tid] = 0.f;
        }
        unsigned long long split[2] = {0ull, 0ull};  // PAIRED: the steps that serve two entries
        const int n = PAIRED   ? build_paired_lists(s, code, tid, wave, lane, split)
                      : HALVES ? build_half_lists(s, code, tid, wave, lane)
                               : build_lists(s, code, tid, wave, lane);
        const int rel_last4 = 4 * (last_contributor - lo);  // slots whose 4 * index is below this are at or before the last contributor

        // (the list is padded with sentinels to a multiple of 4, so a multiple of 2 is always readable)
        // PAIRED: one step per iteration -- the second step's records in flight cost eight registers that this kernel does not have
        // (its lists' base is a per-lane value, the half-wave sums cross a branch); the other waves of the SIMD cover the LDS latency
        constexpr int U = PAIRED ? 1 : 2;
        for (int k = ((n + U - 1) / U) * U - U; k >= 0; k -= U) {
            float4 q0[2], q1[2];
            if (U == 2) {
                load2(s, my_list, k, q0, q1);
            } else {
                const unsigned off = s.list[my_list][k];
                q0[0] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(s.rec) + off);
                q1[0] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(s.rec) + off + 16);
            }
#pragma unroll
            for (int u = U - 1; u >= 0; u--) {
                f2 dxy;
                const float p2 = pair_p2<AM>(q0[u], q1[u], pxy, dxy);
                const float dx = dxy.x, dy = dxy.y;
                const int j4 = __float_as_int(q1[u].z);  // 4 * slot
                // every listed entry was blended by some pixel of this wave (contribution tags): no wave-level tests
                const float oG = alpha_raw<AM, true>(q1[u].y, p2, sb.exptab);  // o G: alpha before the 0.99 clamp, and dalpha/dG * G
                // (the reference tests min(0.99, o G) >= 15/255; 0.99 is above the threshold, so o G itself decides)
                // (written as "not below", so that a NaN o G -- a poisoned opacity -- stays a valid pair with alpha 0.99 as under
                //  the reference's min(0.99f, NaN))
                const bool valid = (j4 < rel_last4) & (p2 <= 0.0f) & !(oG < ALPHA_MIN);
                // No branch: a lane the Gaussian does not reach runs the same instructions with alpha = 0 and o G = 0, which
                // leave its state untouched -- 1/(1 - 0) is exactly 1, so T, and S = 0 X + 1 S, keep their bits -- and
                // make every one of its contributions 0.  (The list's sentinel entries have opacity 0 and an all-zero
                // rgbd entry.)  Valid lanes execute the operations of the reference's order unchanged.  One select (on o G;
                // the clamp of 0 is 0): a select and a compare issue at 4.2 cycles each, a multiply at 2.4.
                const float oGm = valid ? oG : 0.f;
                const float alpha = fminf(0.99f, oGm);
                const float4 cd = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(s.rgbd) + __float_as_int(q1[u].w));
                const float om = 1.f - alpha;
                float inv;
                T = t_div<AM>(T, om, inv);  // backward.cu:570
                const float w = alpha * T;  // dchannel_dcolor = dpixel_depth_ddepth
                const float e = cd.w - gt_px;
                const float X = LEAN ? cd.x * dpix0 + cd.y * dpix1 + cd.z * dpix2 + cd.w * dpix_depth
                                     : cd.x * dpix0 + cd.y * dpix1 + cd.z * dpix2 + cd.w * dpix_depth + (e * e) * dpix_var;
                const float dL_dalpha = (X - S) * T + bg_term * inv;
                S = alpha * X + om * S;  // what the NEXT valid pair (towards the front) subtracts: same operations, same order
                const float qq = oGm * dL_dalpha;  // o G dL/dalpha  (dL_dG * G)
                // Per-lane contributions are the raw moments of qq over the pixel offsets; the factors that depend on the
                // Gaussian only (conic, 1/opacity, the ndc scale) are applied once per (tile, Gaussian) in finish_rows().
                const float qdx = qq * dx, qdy = qq * dy;
                const float wd = w * dpix_depth;
                float tot;
                if (DO_MAP) {
                    // backward.cu:654-664, once per pixel -- at its first valid pair from the back with T > 0.5: the
                    // median-depth term of dL/dmean3D is (v_k - v_{k+1} mul3) * dL/dmedian with factors that depend on
                    // the Gaussian alone, so only the pixel sum of dL/dmedian is formed here, as one more value of the
                    // butterfly (its twelfth slot is free); preprocess_bwd applies the factors.  `mid_thr` turns to +inf
                    // once the pixel has fired.
                    float gmed = 0.f;
                    if (!LEAN) {
                        const bool fire = valid & (T > mid_thr);
                        gmed = fire ? dpix_median : 0.f;
                        mid_thr = fire ? __builtin_inff() : mid_thr;
                    }
                    float g[12];
                    g[0] = w * dpix0;
                    g[1] = w * dpix1;
                    g[2] = w * dpix2;
                    g[3] = LEAN ? wd : wd + 2.f * ((dpix_var * w) * e);  // (2 dL/dvar: the factor is exact wherever it stands)
                    g[4] = qdx;        // sum q dx
                    g[5] = qdy;        // sum q dy
                    g[6] = qdx * dx;   // sum q dx^2
                    g[7] = qdx * dy;   // sum q dx dy
                    g[8] = qdy * dy;   // sum q dy^2
                    g[9] = qq;         // sum q
                    g[10] = DO_POSE ? wd : gmed;   // -> accumulator component 13 / 10
                    g[11] = DO_POSE ? gmed : 0.f;  // -> accumulator component 10
                    if (PAIRED) {
                        float u0, u1;
                        wave_reduce12d_head(g, u0, u1);
                        const int st = k + u;  // (wave-uniform: scalar code)
                        if ((split[st >> 6] >> (st & 63)) & 1ull) {
                            // a pair: each half's own totals to its own entry's column (j4 is uniform in each half)
                            // (the rows are worked out here, one step in nine, rather than kept in registers through the loop)
                            const float r0 = quad_sum(u0), r1 = quad_sum(u1);
                            auto comp_of = [](int slot) { return slot < 10 ? slot : slot == 10 ? (DO_POSE ? 13 : 10) : (DO_POSE ? 10 : -1); };
                            const int s1 = wave_reduce12d_half_slot1(lane);
                            const int c0 = (lane & 3) == 0 ? comp_of(wave_reduce12d_half_slot0(lane)) : -1;
                            const int c1 = ((lane & 3) == 0 && s1 >= 0) ? comp_of(s1) : -1;
                            char* const col = reinterpret_cast<char*>(sb.acc) + j4;
                            if (c0 >= 0) atomicAdd(reinterpret_cast<float*>(col + c0 * (BWD_LD * 4)), r0);
                            if (c1 >= 0) atomicAdd(reinterpret_cast<float*>(col + c1 * (BWD_LD * 4)), r1);
                            continue;
                        }
                        tot = wave_reduce12d_tail(u0, u1);
                    } else {
                        tot = wave_reduce12d(g);
                    }
                } else if (HALVES) {
                    tot = half_reduce3(qdx, qdy, wd);
                } else {
                    float g4[4] = {qdx, qdy, wd, 0.f};
                    tot = wave_reduce4(g4);
                }
                // j4 is wave-uniform here (every lane read the same record; HALVES: uniform in each half-wave)
                if (my_comp >= 0) {
                    float* const cell = reinterpret_cast<float*>(reinterpret_cast<char*>(my_acc) + j4);
                    if (DET) *cell = tot; else atomicAdd(cell, tot);
                }
            }
        }

        __syncthreads();
        if (DET && tid < BWD_NB) {
            // the four planes in wave order into plane 0 (a wave whose tag bit is clear never wrote its column; the sentinel
            // entries of a padded list write column NB, which nobody reads), and the pair's row in the instance-major buffer
            uint32_t row = ~0u;
            if (code != 0u) {
#pragma unroll
                for (int k = 0; k < NACC_LIGHT; k++) {
                    // (components this variant's lanes deliver -- my_comp above; the others are nobody's and read as zero)
                    const bool delivered = DO_MAP ? (k <= 9 || k == 10 || (DO_POSE && k == 13)) : (k == 4 || k == 5 || k == 13);
                    float v = 0.f;
                    if (delivered) {
#pragma unroll
                        for (int w = 0; w < 4; w++)
                            if ((code >> w) & 1u) v += sb.acc[w * SB::PLANE + k * BWD_LD + tid];
                    }
                    sb.acc[k * BWD_LD + tid] = v;
                }
                const uint32_t gid = s.id[tid];
                const ushort4 rc = a.det_rect[gid];
                row = a.det_goff[gid] + (uint32_t)(ty - (int)rc.y) * (uint32_t)(rc.z - rc.x) + (uint32_t)(tx - (int)rc.x);
            }
            sb.inst[tid] = row;
        }
        if (DET) __syncthreads();
        // moments -> gradients, one thread per staged Gaussian (backward.cu:627-631, 669-678):
        //   dL/dmean2D = -(a Sx + b Sy) W/2, -(c Sy + b Sx) H/2;  dL/dconic = -Sxx/2, -Sxy/2, -Syy/2;  dL/dopacity = S0/o
        if (code != 0u) {
            constexpr float UN = AlphaPath<AM>::PUNSCALE;  // (undoes the scale of the staged conic)
            const float4 r0 = s.rec[2 * tid], r1 = s.rec[2 * tid + 1];
            const float ca = r0.z * (-2.f * UN), cb = r1.x * (-UN), cc = r0.w * (-2.f * UN);  // unscaled conic
            const float Sx = sb.acc[4 * BWD_LD + tid], Sy = sb.acc[5 * BWD_LD + tid];
            sb.acc[4 * BWD_LD + tid] = -(ca * Sx + cb * Sy) * ddelx_dx;
            sb.acc[5 * BWD_LD + tid] = -(cc * Sy + cb * Sx) * ddely_dy;
            if (DO_MAP) {
                sb.acc[6 * BWD_LD + tid] *= -0.5f;
                sb.acc[7 * BWD_LD + tid] *= -0.5f;
                sb.acc[8 * BWD_LD + tid] *= -0.5f;
                sb.acc[9 * BWD_LD + tid] *= __builtin_amdgcn_rcpf(r1.y);
            }
        }
        __syncthreads();
        if (DET) {  // 16 consecutive lanes store one pair's 64-byte row (components 14, 15 stay zero)
            const int comp = tid & 15;
            for (int r = tid >> 4; r < cnt; r += 16) {
                const uint32_t row = sb.inst[r];
                if (row < a.det_R && comp < NACC_LIGHT) a.det_rows[(size_t)row * DGR_ACC_STRIDE + comp] = sb.acc[comp * BWD_LD + r];
            }
        } else {
            flush_acc<NACC_LIGHT, BWD_LD>(sb.acc, s.id, cnt, a.acc, tid);
        }
    }
}

// BY_FRAME: half-wave / paired lists or quadrant lists by the frame's flag (render_fwd_light_kernel above: the forward that wrote the
// tags took the same branch -- a quadrant-list forward sets both halves' bits of a quadrant, which a half-wave backward could walk
// correctly but to no gain).  Without it: quadrant lists, for which the tag bytes of either forward are folded to four bits -- the
// deterministic kernels (LDS for the planes) and the glibc form (an A/B mode that spills a register with eight lists).
template <int AM, bool DO_MAP, bool DO_POSE, bool DET = false, bool LEAN = false, bool BY_FRAME = false>
__global__ void __launch_bounds__(256, 8) render_bwd_light_kernel(RenderBwdLightArgs a) {
    static_assert(!BY_FRAME || !DET, "the deterministic kernel walks quadrant lists");
    __shared__ union { StagedBwd<DET, BY_FRAME> h; StagedBwd<DET, false> q; } sb;
    bool quadrant_lists = true;
    const uint4 slot = blend_slot(a.sched, a.ranges, a.sched_flag, a.grid_x * a.grid_y, nullptr, BY_FRAME ? &quadrant_lists : nullptr);
    if (BY_FRAME && !quadrant_lists) render_bwd_light_body<AM, DO_MAP, DO_POSE, DET, LEAN, BY_FRAME>(a, sb.h, slot);
    else render_bwd_light_body<AM, DO_MAP, DO_POSE, DET, LEAN, false>(a, sb.q, slot);
}

// ---- deterministic gradients: the kernels around the DET blend backward
// n = tiles of a Gaussian's rectangle; goff = exclusive prefix of n over the Gaussians (the reference's point_offsets,
// L/cuda_rasterizer/rasterizer_impl.cu:283, which the segment binning never needs): per-block sums, then per-block bases.
__device__ __forceinline__ uint32_t rect_tiles(ushort4 r) { return (r.z > r.x && r.w > r.y) ? (uint32_t)(r.z - r.x) * (uint32_t)(r.w - r.y) : 0u; }
__global__ void __launch_bounds__(256) det_block_sums_kernel(int P, const ushort4* __restrict__ rect, uint32_t* __restrict__ blk) {
    __shared__ uint32_t red[4];
    const int g = blockIdx.x * 256 + threadIdx.x;
    uint32_t v = g < P ? rect_tiles(rect[g]) : 0u;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) blk[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ void __launch_bounds__(256) det_offsets_kernel(int P, const ushort4* __restrict__ rect, const uint32_t* __restrict__ blk,
                                                          uint32_t* __restrict__ goff) {
    __shared__ uint32_t red[4], wsum[4];
    uint32_t before = 0;  // instances of the blocks in front of this one (integer sums: any order)
    for (int b = threadIdx.x; b < (int)blockIdx.x; b += 256) before += blk[b];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) before += __shfl_xor(before, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = before;
    const int g = blockIdx.x * 256 + threadIdx.x;
    const uint32_t n = g < P ? rect_tiles(rect[g]) : 0u;
    uint32_t incl = n;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t v = __shfl_up(incl, off, 64);
        if ((threadIdx.x & 63) >= off) incl += v;
    }
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t base = red[0] + red[1] + red[2] + red[3];
    for (int w = 0; w < (int)(threadIdx.x >> 6); w++) base += wsum[w];
    if (g < P) goff[g] = base + incl - n;
}
// a Gaussian's rows, contiguous from goff, added in ascending order: 16 lanes per Gaussian, a lane per component
__global__ void __launch_bounds__(256) det_gather_kernel(int P, const ushort4* __restrict__ rect, const uint32_t* __restrict__ goff,
                                                         const float* __restrict__ rows, uint32_t R, float* __restrict__ acc) {
    const int g = blockIdx.x * 16 + (threadIdx.x >> 4), comp = threadIdx.x & 15;
    if (g >= P) return;
    uint32_t n = rect_tiles(rect[g]);
    if (n == 0u) return;
    const uint32_t first = goff[g];
    // R (the caller's: a lazy forward's capacity, or a wrong value) smaller than the rectangles' total: this Gaussian's rows were
    // not all written -- a truncated sum would be a plausible gradient, so the row is NaN instead (and every gradient of the
    // Gaussian with it: preprocess_bwd reads the row)
    if (first >= R || n > R - first) {
        acc[(size_t)g * DGR_ACC_STRIDE + comp] = __builtin_nanf("");
        return;
    }
    const float* r = rows + (size_t)first * DGR_ACC_STRIDE + comp;
    float v = 0.f;
    for (uint32_t i = 0; i < n; i++) v += r[(size_t)i * DGR_ACC_STRIDE];
    acc[(size_t)g * DGR_ACC_STRIDE + comp] = v;
}

// self-test of the butterflies: in[c * 64 + lane] -> the three networks' results and value maps per lane (dgr_debug_wave_reduce)
__global__ void __launch_bounds__(64) wave_reduce_test_kernel(const float* in, float* out16, float* out12, float* out4,
                                                             int* comp16, int* comp12, int* comp4) {
    const int lane = threadIdx.x;
    float g16[16], g12[12], g4[4];
#pragma unroll
    for (int k = 0; k < 16; k++) g16[k] = in[k * 64 + lane];
#pragma unroll
    for (int k = 0; k < 12; k++) g12[k] = g16[k];
#pragma unroll
    for (int k = 0; k < 4; k++) g4[k] = g16[k];
    out16[lane] = wave_reduce16d(g16);
    out12[lane] = wave_reduce12d(g12);
    out4[lane] = wave_reduce4(g4);
    comp16[lane] = wave_reduce16d_comp(lane);
    comp12[lane] = wave_reduce12d_comp(lane);
    comp4[lane] = wave_reduce4_comp(lane);
}

// self-test of the reductions per HALF of the wave (wave_reduce.h): in[c * 64 + lane], twelve values for the paired step of the
// mapping backward (r0, r1 and the butterfly slot each lane holds), the first three for the tracking backward's half_reduce3
__global__ void __launch_bounds__(64) half_reduce_test_kernel(const float* in, float* r0, float* r1, float* h3, int* slot0, int* slot1,
                                                             int* comp3) {
    const int lane = threadIdx.x;
    float g[12];
#pragma unroll
    for (int k = 0; k < 12; k++) g[k] = in[k * 64 + lane];
    h3[lane] = half_reduce3(g[0], g[1], g[2]);
    float u0, u1;
    wave_reduce12d_head(g, u0, u1);
    r0[lane] = quad_sum(u0);
    r1[lane] = quad_sum(u1);
    slot0[lane] = wave_reduce12d_half_slot0(lane);
    slot1[lane] = wave_reduce12d_half_slot1(lane);
    comp3[lane] = half_reduce3_comp(lane);
}

// ... and of the sixteen-value network stopped before its cross-half stage (the paired step of the FULL backward): in[c * 64 + lane]
__global__ void __launch_bounds__(64) half_reduce16_test_kernel(const float* in, float* r0, float* r1, int* slot0, int* slot1) {
    const int lane = threadIdx.x;
    float g[16];
#pragma unroll
    for (int k = 0; k < 16; k++) g[k] = in[k * 64 + lane];
    float u0, u1;
    wave_reduce16d_head(g, u0, u1);
    r0[lane] = quad_sum(u0);
    r1[lane] = quad_sum(u1);
    slot0[lane] = wave_reduce16d_half_slot0(lane);
    slot1[lane] = wave_reduce16d_half_slot1(lane);
}

// self-test of the list builders of render_common.h on one batch of 128 staged slots: codes[slot] = the eight bits "half h of
// quadrant wave w" (bit 2 w + h).  paired[w] / halves[w] (280 words each) = {steps or length, split[0] lo, hi, split[1] lo, hi,
// list 2 w [0..135], list 2 w + 1 [0..135]} as wave w leaves them (dgr_debug_lane_lists).
__global__ void __launch_bounds__(256) lane_lists_test_kernel(const unsigned char* codes, uint32_t* paired, uint32_t* halves) {
    typedef StagedT<128, uint32_t, 8> S;
    __shared__ S s;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const unsigned code = tid < 128 ? codes[tid] : 0u;
    for (int pass = 0; pass < 2; pass++) {
        for (int i = tid; i < 8 * S::LIST_LD; i += 256) (&s.list[0][0])[i] = 0xFFFFFFFFu;
        __syncthreads();
        unsigned long long split[2] = {0ull, 0ull};
        const int n = pass == 0 ? build_paired_lists(s, code, tid, wave, lane, split) : build_half_lists(s, code, tid, wave, lane);
        uint32_t* const o = (pass == 0 ? paired : halves) + 280 * wave;
        if (lane == 0) {
            o[0] = (uint32_t)n;
            o[1] = (uint32_t)split[0]; o[2] = (uint32_t)(split[0] >> 32);
            o[3] = (uint32_t)split[1]; o[4] = (uint32_t)(split[1] >> 32);
        }
        for (int i = lane; i < S::LIST_LD; i += 64) {
            o[5 + i] = s.list[2 * wave][i];
            o[5 + S::LIST_LD + i] = s.list[2 * wave + 1][i];
        }
        __syncthreads();
    }
}

// self-test of exact_math.h: out_exp[i] = exp_p32(x[i]) (GLIBC: exp_glibc), out_div[i] = div_ref(a[i], b[i]) (dgr_debug_exact_math)
template <bool GLIBC>
__global__ void __launch_bounds__(256) exact_math_test_kernel(int n, const float* x, const float* a, const float* b, float* out_exp,
                                                             float* out_div) {
    __shared__ uint64_t tab[32];
    exp_ref_table_fill(tab, threadIdx.x);
    __syncthreads();
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        // (the clamped forms: equal to the plain ones down to -104, 0 below)
        out_exp[i] = GLIBC ? exp_glibc<true>(x[i], tab) : exp_p32<true>(x[i]);
        float inv;
        out_div[i] = t_div<ALPHA_REF>(a[i], b[i], inv);
    }
}

template <int AM>
void launch_bwd_light_mode(const RenderBwdLightArgs& a, int tiles, hipStream_t stream) {
    // (the lane lists -- half-wave / paired or per quadrant -- are the frame's own choice: render_bwd_light_kernel)
    if (AM == ALPHA_REF && !a.det_rows && !a.dL_dpix_median && !a.dL_dpix_var) {
        if (!a.map_off && !a.track_off)
            launch_blend((render_bwd_light_kernel<ALPHA_REF, true, true, false, true, true>), dim3(tiles), dim3(256), stream, a);
        else if (!a.map_off)
            launch_blend((render_bwd_light_kernel<ALPHA_REF, true, false, false, true, true>), dim3(tiles), dim3(256), stream, a);
        else
            launch_blend((render_bwd_light_kernel<ALPHA_REF, false, true, false, true, true>), dim3(tiles), dim3(256), stream, a);
        return;
    }
    if (AM == ALPHA_REF && a.det_rows) {
        if (!a.map_off && !a.track_off)
            launch_blend((render_bwd_light_kernel<ALPHA_REF, true, true, true>), dim3(tiles), dim3(256), stream, a);
        else if (!a.map_off)
            launch_blend((render_bwd_light_kernel<ALPHA_REF, true, false, true>), dim3(tiles), dim3(256), stream, a);
        else
            launch_blend((render_bwd_light_kernel<ALPHA_REF, false, true, true>), dim3(tiles), dim3(256), stream, a);
        return;
    }
    constexpr bool BF = AM != ALPHA_GLIBC;  // (the glibc form, an A/B mode, spills a register with the eight lists: quadrant lists)
    if (!a.map_off && !a.track_off)
        launch_blend((render_bwd_light_kernel<AM, true, true, false, false, BF>), dim3(tiles), dim3(256), stream, a);
    else if (!a.map_off)
        launch_blend((render_bwd_light_kernel<AM, true, false, false, false, BF>), dim3(tiles), dim3(256), stream, a);
    else
        launch_blend((render_bwd_light_kernel<AM, false, true, false, false, BF>), dim3(tiles), dim3(256), stream, a);
}
}  // namespace

hipError_t launch_render_fwd_light(const RenderFwdLightArgs& a, int alpha_mode, hipStream_t stream) {
    const int tiles = a.grid_x * a.grid_y;
    if (tiles <= 0) return hipSuccess;
    switch (alpha_mode) {
        case ALPHA_FAST: launch_blend(render_fwd_light_kernel<ALPHA_FAST>, dim3(tiles), dim3(256), stream, a); break;
        case ALPHA_GLIBC: launch_blend(render_fwd_light_kernel<ALPHA_GLIBC>, dim3(tiles), dim3(256), stream, a); break;
        default: launch_blend(render_fwd_light_kernel<ALPHA_REF>, dim3(tiles), dim3(256), stream, a);
    }
    return hipGetLastError();
}
hipError_t launch_render_bwd_light(const RenderBwdLightArgs& a, int alpha_mode, hipStream_t stream) {
    const int tiles = a.grid_x * a.grid_y;
    if (tiles <= 0 || (a.track_off && a.map_off)) return hipSuccess;
    switch (alpha_mode) {
        case ALPHA_FAST: launch_bwd_light_mode<ALPHA_FAST>(a, tiles, stream); break;
        case ALPHA_GLIBC: launch_bwd_light_mode<ALPHA_GLIBC>(a, tiles, stream); break;
        default: launch_bwd_light_mode<ALPHA_REF>(a, tiles, stream);
    }
    return hipGetLastError();
}
hipError_t launch_det_offsets(int P, const ushort4* rect, uint32_t* blk, uint32_t* goff, hipStream_t stream) {
    if (P <= 0) return hipSuccess;
    const int nb = (P + 255) / 256;
    launch(det_block_sums_kernel, dim3(nb), dim3(256), stream, P, rect, blk);
    launch(det_offsets_kernel, dim3(nb), dim3(256), stream, P, rect, (const uint32_t*)blk, goff);
    return hipGetLastError();
}
hipError_t launch_det_gather(int P, const ushort4* rect, const uint32_t* goff, const float* rows, uint32_t R, float* acc, hipStream_t stream) {
    if (P <= 0) return hipSuccess;
    launch(det_gather_kernel, dim3((P + 15) / 16), dim3(256), stream, P, rect, goff, rows, R, acc);
    return hipGetLastError();
}
hipError_t launch_exact_math_test(int n, const float* x, const float* a, const float* b, float* out_exp, float* out_div,
                                  int alpha_mode, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    if (alpha_mode == ALPHA_GLIBC)
        launch(exact_math_test_kernel<true>, dim3(min((n + 255) / 256, 4096)), dim3(256), stream, n, x, a, b, out_exp, out_div);
    else
        launch(exact_math_test_kernel<false>, dim3(min((n + 255) / 256, 4096)), dim3(256), stream, n, x, a, b, out_exp, out_div);
    return hipGetLastError();
}
hipError_t launch_wave_reduce_test(const float* in, float* out16, float* out12, float* out4, int* comp16, int* comp12, int* comp4,
                                   hipStream_t stream) {
    launch(wave_reduce_test_kernel, dim3(1), dim3(64), stream, in, out16, out12, out4, comp16, comp12, comp4);
    return hipGetLastError();
}

hipError_t launch_half_reduce_test(const float* in, float* r0, float* r1, float* h3, int* slot0, int* slot1, int* comp3, hipStream_t stream) {
    launch(half_reduce_test_kernel, dim3(1), dim3(64), stream, in, r0, r1, h3, slot0, slot1, comp3);
    return hipGetLastError();
}
hipError_t launch_half_reduce16_test(const float* in, float* r0, float* r1, int* slot0, int* slot1, hipStream_t stream) {
    launch(half_reduce16_test_kernel, dim3(1), dim3(64), stream, in, r0, r1, slot0, slot1);
    return hipGetLastError();
}
hipError_t launch_lane_lists_test(const unsigned char* codes, uint32_t* paired, uint32_t* halves, hipStream_t stream) {
    launch(lane_lists_test_kernel, dim3(1), dim3(256), stream, codes, paired, halves);
    return hipGetLastError();
}

}  // namespace dgr
