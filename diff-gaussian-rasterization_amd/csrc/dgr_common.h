// dgr_common.h -- state-buffer layouts and small device helpers shared by the gfx950 kernels.
//
// The reference carves GeometryState / BinningState / ImageState out of three byte buffers
// (L/cuda_rasterizer/rasterizer_impl.h:29-72, rasterizer_impl.cu:155-193).  The buffers stay
// opaque at the boundary, so the layout here is chosen for the MI355X kernels instead:
//   geometry : one 48-byte render record per Gaussian in a 64-byte slot (3 x float4, gathered as whole 16-B
//              pieces by the blend kernels, from ONE L2 line), depth, radius, tile rect (4 x u16), clamp bits (the 3D covariance is
//              NOT kept: the backward re-forms it from scale and rotation, same expression, same bits)
//   image    : per-tile {count, fill, range} + per-pixel n_contrib (+ full: final T, n_valid)
//   binning  : per-instance 64-bit sort keys (depth bits << 32 | gaussian id), the sorted id list, arrival ranks
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DGR_BLOCK_X 16  // cuda_rasterizer/config.h:16-17 -- fixes the tile-key layout
#define DGR_BLOCK_Y 16
#define DGR_TILE_PIX 256
#define DGR_NEAR 0.2f  // cuda_rasterizer/auxiliary.h:152
// point_list entries: Gaussian id in the low 28 bits, the light forward's contribution tag in the top 4
#define DGR_TAG_SHIFT 28
#define DGR_ID_MASK ((1u << DGR_TAG_SHIFT) - 1u)
#ifndef DGR_COUNT_STRIDE
#define DGR_COUNT_STRIDE 16
#endif  // tile counters are padded to one per 64-byte line: returning atomics on one line serialise

#define DGR_SCHED_CLASSES 32

namespace dgr {

__host__ __device__ inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// list length -> class of the blend kernels' tile schedule (binning.hip): 0 for an empty list, then two classes per octave
// (lengths within ~40 % of each other share a class), saturating at 31 (>= 23 170 entries).  Monotone in n.
__device__ inline uint32_t sched_class(uint32_t n) {
    if (n == 0u) return 0u;
    const uint32_t l = 31u - (uint32_t)__clz((int)n);
    const uint32_t half = l ? (n >> (l - 1u)) & 1u : 0u;
    const uint32_t c = 1u + 2u * l + half;
    return c < (uint32_t)DGR_SCHED_CLASSES - 1u ? c : (uint32_t)DGR_SCHED_CLASSES - 1u;
}

// ---- render record ---------------------------------------------------------------------------
// q0 = {x_pix, y_pix, depth, opacity}   (means2D, depths, conic_opacity.w of the reference)
// q1 = {conic a, conic b, conic c, 0}
// q2 = {r, g, b, 0}                     (geomState.rgb, or a copy of colors_precomp)
// 48 bytes of data in a 64-byte slot (DGR_REC_STRIDE float4s): the blend kernels gather one record per list entry, the L2 fetches
// whole 64-byte lines (TCC_EA0_RDREQ_32B = 0 on gfx950), and a record at a 48-byte stride lies across two lines half of the time
// -- 1.5 lines per gather against exactly one (profiles/r9/fwd_traffic.txt).  The fourth float4 is never written or read.
#ifndef DGR_REC_STRIDE
#define DGR_REC_STRIDE 4
#endif
struct GeometryView {
    float4* rec;        // [DGR_REC_STRIDE * P]
    float* depths;      // [P]
    int* radii;         // [P] internal copy (the caller's `radii` may be NULL)
    ushort4* rect;      // [P] {xmin, ymin, xmax, ymax} in tiles; all-zero when culled
    uint8_t* clamped;   // [P] bit c set <=> channel c was clamped at 0
    uint32_t* goff;     // [P] index of the Gaussian's first instance in Gaussian-major order (the reference's
                        //     point_offsets, exclusive form); written by count_rank
    uint32_t* block_tiles;  // [ceil(P/256)] instances produced by each 256-Gaussian block of preprocess; turned into its
                            //     exclusive prefix in place by scan_blocks
    float4* shd;        // [3][P] (three planes) d(colour)/d(view direction x, y, z) of the SH evaluation ({r, g, b, -} each), written by
                        //     preprocess_fwd for the Gaussians whose colour it evaluates: the backward needs the 45 higher SH
                        //     coefficients only through these nine numbers, so it reads 48 bytes instead of the 192-byte row
    size_t bytes;
};
__host__ __device__ inline GeometryView carve_geometry(char* base, int P) {
    GeometryView g;
    size_t o = 0;
    g.rec = (float4*)(base + o);      o = align_up(o + sizeof(float4) * DGR_REC_STRIDE * (size_t)P, 256);
    g.depths = (float*)(base + o);    o = align_up(o + sizeof(float) * (size_t)P, 256);
    g.radii = (int*)(base + o);       o = align_up(o + sizeof(int) * (size_t)P, 256);
    g.rect = (ushort4*)(base + o);    o = align_up(o + sizeof(ushort4) * (size_t)P, 256);
    g.clamped = (uint8_t*)(base + o); o = align_up(o + (size_t)P, 256);
    g.goff = (uint32_t*)(base + o);   o = align_up(o + sizeof(uint32_t) * (size_t)P, 256);
    g.block_tiles = (uint32_t*)(base + o); o = align_up(o + sizeof(uint32_t) * (((size_t)P + 255) / 256), 256);
    g.shd = (float4*)(base + o);      o = align_up(o + sizeof(float4) * 3 * (size_t)P, 256);
    g.bytes = o;
    return g;
}

// what the forward asks of the binning kernel that writes ImageView::cursor[3] (launch_bin_tiles / launch_scan_tiles: blend_flags)
enum { BLEND_SCHEDULE = 1, BLEND_LISTS_QUADRANT = 4, BLEND_LISTS_AUTO = 8 };

struct ImageView {
    int* status;          // [4] {num_rendered, overflow, prefiltered violation, reserved}
    uint32_t* cursor;     // [4] presized path: {instances handed out so far (preprocess_fwd's blocks bump it to place their
                          //     ranks), prefiltered violation seen}; cleared together with the tile counters.
                          //     [2] = capacity the binning buffer was carved with (scan_tiles / bin_tiles)
                          //     [3] = the frame's blend flags: bit 0: tile_sched holds this frame's schedule (0: the blend kernels use the
                          //           static band map); bit 1: the binning buffer overflowed; bit 2: the light blend kernels walk
                          //           quadrant lists rather than half-wave lists (BLEND_* below: what the host asks the binning for)
    uint32_t* tile_count; // [tiles * DGR_COUNT_STRIDE] instances per tile (histogram filled by count_rank), one per line
    uint2* ranges;        // [tiles] {start, end} into point_list
    uint4* tile_sched;    // [tiles] the blend kernels' schedule: workgroup b works on tile .x, whose list is [.y, .z) --
                          //     classes of long lists first, neighbours on one XCD (tile_schedule_kernel, binning.hip)
    uint32_t* n_contrib;  // [N]
    float* final_T;       // [N]   (full variant)
    uint32_t* n_valid;    // [N]   (full variant) valid contributors of the pixel
    uint32_t* first_contrib;  // [N] (full variant) 1-based list position of the pixel's first valid contributor, 0 = none
    size_t bytes;
};
__host__ __device__ inline int tiles_x(int W) { return (W + DGR_BLOCK_X - 1) / DGR_BLOCK_X; }
__host__ __device__ inline int tiles_y(int H) { return (H + DGR_BLOCK_Y - 1) / DGR_BLOCK_Y; }
__host__ __device__ inline ImageView carve_image(char* base, int W, int H) {
    ImageView v;
    const size_t tiles = (size_t)tiles_x(W) * tiles_y(H), N = (size_t)W * H;
    size_t o = 0;
    v.status = (int*)(base + o);
    v.cursor = (uint32_t*)(base + o + 64);  o = align_up(o + 4 * sizeof(int), 256);
    v.tile_count = (uint32_t*)(base + o); o = align_up(o + tiles * 4 * DGR_COUNT_STRIDE, 256);
    v.ranges = (uint2*)(base + o);        o = align_up(o + tiles * 8, 256);
    v.tile_sched = (uint4*)(base + o);    o = align_up(o + tiles * 16, 256);
    v.n_contrib = (uint32_t*)(base + o);  o = align_up(o + N * 4, 256);
    v.final_T = (float*)(base + o);       o = align_up(o + N * 4, 256);
    v.n_valid = (uint32_t*)(base + o);    o = align_up(o + N * 4, 256);
    v.first_contrib = (uint32_t*)(base + o); o = align_up(o + N * 4, 256);
    v.bytes = o;
    return v;
}

// bijective XCD-aware remap (workgroup b runs on XCD b % 8): XCD x gets a contiguous run of the n items
__host__ __device__ inline int xcd_contiguous(int b, int n) {
    const int xcd = b & 7, local = b >> 3;
    const int q = n >> 3, r = n & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + local;
}

// Where a forward reports its status word when a status slot is armed (api.hip: dgr_status_arm): `host` = pinned host memory
// mapped into the device's address space.  The word is complete when the binning kernels have finished, so the first workgroup
// of the forward blend -- the next kernel on the stream -- copies it there: {num_rendered, overflow, prefiltered violation, 0,
// tag, longest tile list}, `tag` last, and the host reads it without a copy, an event or a wait.  `ws` = a device word owned by
// the slot, zero between forwards: the binning kernels leave the frame's longest tile list in it (an atomic max per segment).
struct StatusReport {
    int* host;      // NULL: no report
    uint32_t tag;
    uint32_t* ws;
};
// (one thread of the forward blend, before its workgroup does anything else; `status` = the frame's device status word)
__device__ __forceinline__ void report_status(const StatusReport& r, const int* status) {
    const int total = __hip_atomic_load(status + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int overflow = __hip_atomic_load(status + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int flag = __hip_atomic_load(status + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t longest = __hip_atomic_load(r.ws, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(r.ws, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(r.host + 0, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(r.host + 1, overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(r.host + 2, flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(r.host + 3, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(r.host + 5, (int)longest, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    // the tag announces the word: a system-scope RELEASE store (the host side reads the tag with an acquire load); the slot is
    // allocated hipHostMallocMapped | hipHostMallocCoherent, so nothing here rests on the default coherence of host memory
    __hip_atomic_store(r.host + 4, (int)r.tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

struct BinningView {
    uint32_t* point_list;  // [cap] sorted gaussian ids (binningState.point_list); at offset 0 so that the
                           //       backward can find it without knowing the capacity
    uint64_t* keys;        // [cap] (depth bits << 32 | gaussian id), grouped by tile, unsorted (sort_tiles reads them)
    uint32_t* ranks;       // [cap] Gaussian-major: arrival rank of each instance within its tile (count_rank -> emit)
    // segment binning (segment_binning.hip): one PAIR per (Gaussian, tile row, 16-tile segment) it touches
    uint64_t* pair_keys;   // [cap] (depth bits << 32 | gaussian id), grouped by producing workgroup, then by segment
    uint8_t* pair_cov;     // [cap] columns covered inside the segment: first | last << 4  (shares the bytes of `ranks`:
                           //       a forward uses either the global-atomic count or the segment binning).  CLOBBERED by the
                           //       light forward's blend, which keeps its contribution tags per half of a quadrant in these
                           //       bytes for the backward (render_common.h: half_tags / stage_tagged): part of a view's state
    size_t bytes;
};
__host__ __device__ inline BinningView carve_binning(char* base, size_t cap) {
    BinningView b;
    size_t o = 0;
    b.point_list = (uint32_t*)(base + o); o = align_up(o + 4 * cap, 256);
    b.keys = (uint64_t*)(base + o);       o = align_up(o + 8 * cap, 256);
    b.ranks = (uint32_t*)(base + o);      b.pair_cov = (uint8_t*)(base + o); o = align_up(o + 4 * cap, 256);
    b.pair_keys = (uint64_t*)(base + o);  o = align_up(o + 8 * cap, 256);
    b.bytes = o;
    return b;
}

// Forward-only workspace of the segment binning (segment_binning.hip), carved behind the binning arrays.  A tile row is cut
// into segments of 16, 8 or 4 tiles (chosen per call; the tables are carved for 4); workgroup w of bin_segments (at most
// SEG_MAX_WGS) leaves
//   pair_off[s][w]  start of its run of pairs for segment s (row nseg: end of its region; row nseg + 1: how many of its Gaussians
//                   are on screen -- the frame's instances per visible Gaussian decide the blend kernels' lane lists), and
//   inst_pre[s][w]  its instances in the segments 0..s (a running sum): the column sums are the list starts.
#define SEG_TILES_MAX 16
#define SEG_TILES_MIN 4
#define SEG_MAX_WGS 256
#define SEG_K1_LDS_MAX (128 * 1024)  // bin_segments holds 16 bytes per segment in LDS
struct SegmentTables {
    uint32_t* pair_off;  // [(nseg + 2) * SEG_MAX_WGS]
    uint32_t* inst_pre;  // [nseg * SEG_MAX_WGS]
    size_t bytes;
};
__host__ __device__ inline SegmentTables carve_segment_tables(char* base, int W, int H) {
    SegmentTables t;
    const size_t gx = (size_t)((W + DGR_BLOCK_X - 1) / DGR_BLOCK_X), gy = (size_t)((H + DGR_BLOCK_Y - 1) / DGR_BLOCK_Y);
    const size_t nseg = gy * ((gx + SEG_TILES_MIN - 1) / SEG_TILES_MIN);
    size_t o = 0;
    if (gx == 0 || gy == 0 || nseg * 16 > SEG_K1_LDS_MAX) { t.pair_off = nullptr; t.inst_pre = nullptr; t.bytes = 0; return t; }
    t.pair_off = (uint32_t*)(base + o); o = align_up(o + 4 * (nseg + 2) * SEG_MAX_WGS, 256);
    t.inst_pre = (uint32_t*)(base + o); o = align_up(o + 4 * nseg * SEG_MAX_WGS, 256);
    t.bytes = o;
    return t;
}

// ---- backward scratch --------------------------------------------------------------------------
// One 64-byte accumulator row per Gaussian so that every atomic of a (pixel, Gaussian) pair lands
// in a single cache line:
//  [0..2] dL/dcolour  [3] dL/ddepth (blend + variance terms)  [4..5] dL/dmean2D
//  [6..8] dL/dconic (xx, xy, yy)  [9] dL/dopacity  [10] sum of dL/dmedian over the pixels whose median this is  [11..12] unused
//  [13] sum of the blend-only depth term (pose gradient)  [14..15] unused
// full variant: [0..9] as above ([3] = dL/dgau_depth incl. the uncertainty term), then the sums its pose gradient
// (ComputePG, F/cuda_rasterizer/backward.cu:838-1338) is linear in:
//  [10..11] colour-only dL/d(ndc x, y)   [12] dL_depth * alpha T of the pixels whose FRONT-MOST valid Gaussian this is
//  [13..14] dL_depth * d(depth)/d(ndc x, y) of those same pixels   [15] unused
#define DGR_ACC_STRIDE 16
#define DGR_POSE_BUCKETS 64
struct BackwardScratch {
    float* acc;         // [P * 16]
    uint32_t* ticket;   // [1] blocks of preprocess_bwd that have delivered their pose partial (the last one finishes the sum)
    double* pose_part;  // [DGR_POSE_BUCKETS * 12] partial sums of the pose gradient, block b adds into bucket b % 64
    size_t zero_bytes;  // acc, ticket and pose_part are contiguous and cleared together before the blend backward
    size_t bytes;
};
__host__ __device__ inline BackwardScratch carve_backward_scratch(char* base, int P) {
    BackwardScratch s;
    size_t o = 0;
    s.acc = (float*)(base + o);         o = align_up(o + sizeof(float) * DGR_ACC_STRIDE * (size_t)P, 256);
    s.ticket = (uint32_t*)(base + o);   o = align_up(o + 4, 256);
    s.pose_part = (double*)(base + o);  o = align_up(o + sizeof(double) * 12 * DGR_POSE_BUCKETS, 256);
    s.zero_bytes = o;
    s.bytes = o;
    return s;
}

}  // namespace dgr
