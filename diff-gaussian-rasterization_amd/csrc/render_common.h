// render_common.h -- staging and traversal helpers shared by the light and full blend kernels (gfx950).
// See the header comment of render_light.hip for the design (per-wave compacted record lists, quadrant culling).
#pragma once
#include "dgr_common.h"
#include "kernels.h"
#include "wave_reduce.h"
#include "exact_math.h"

namespace dgr {

// Workgroup b of a blend kernel -> {tile, list start, list end}.  With a schedule (tile_schedule_kernel, binning.hip: classes of
// long lists first, neighbours on one XCD) that is one 16-byte entry.  A frame whose lists are even needs none -- the schedule
// then is the static map with extra steps: a launch, 11 us in front of the blend at 1080p, and neighbours spread a little --
// so the forward skips the kernel (api.hip: the policy) and the workgroups fall back on the band map of rounds 1-5: workgroup
// b runs on XCD b % 8, every XCD takes a contiguous band of the image (neighbouring tiles share the records in its L2).
// Which of the two a frame uses is recorded in its image state by the binning kernel (cursor[3]) for forward and backward alike.
// The same word's bit 1 says that the frame's binning buffer OVERFLOWED (every tile list is empty): the forward kernels then write NaN
// images instead of a plausible empty frame (a lazy caller learns of the overflow a call or two later).  It rides in this word
// Bit 2: the frame's lane lists (light blend kernels): one per quadrant wave instead of one per half-wave -- the binning kernel decides
// it per frame (segment_binning.hip) and forward and backward of the frame take the same uniform branch on it.
// The flags ride in this word because the status word itself is no place to read from here: the full forward's workgroups add their valid-pair counts to
// status[3] with atomics as they finish, and a load of that line queues behind them -- 34 -> 60 us for render_fwd at config 2.
__device__ __forceinline__ uint4 blend_slot(const uint4* __restrict__ sched, const uint2* __restrict__ ranges,
                                            const uint32_t* __restrict__ sched_flag, int tiles, bool* overflowed = nullptr,
                                            bool* quadrant_lists = nullptr) {
    const int flag = __builtin_amdgcn_readfirstlane((int)*sched_flag);
    if (overflowed) *overflowed = (flag & 2) != 0;
    if (quadrant_lists) *quadrant_lists = (flag & 4) != 0;
    if (flag & 1) return sched[blockIdx.x];
    const int tile = xcd_contiguous((int)blockIdx.x, tiles);
    const uint2 r = ranges[tile];
    return make_uint4((uint32_t)tile, r.x, r.y, 0u);
}

// wave-uniform "any lane": one v_cmp into an SGPR pair + s_cmp (HIP's __any() goes through v_cndmask + v_cmp)
__device__ __forceinline__ bool wave_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }

namespace {

constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
constexpr float ALPHA_MIN = 15.0f / 255.0f;  // forward.cu:365

// How the blend kernels evaluate alpha = min(0.99, o exp(power)) and T / (1 - alpha)  (exact_math.h):
//   ALPHA_REF  (default): the reference's expression in the reference's association (forward.cu:354-364,
//       backward.cu:561-570), expf with the CPU restatement's bits (exp_p32: an fp32-only polynomial expf, <= 0.9 ulp, the
//       same operation sequence in oracle/dgr_oracle.cpp), correctly rounded division (div_ref).  The light backward
//       amplifies a last-bit difference of one alpha by 1 / T_final and by alpha / (1 - alpha) per division, so agreement
//       with the CPU restatement to 1e-5 needs the same BITS, not merely the same accuracy.
//   ALPHA_GLIBC (dgr_set_option("alpha_mode", 2)): as ALPHA_REF with glibc's expf algorithm in the double pipe (exp_glibc) --
//       rounds 5-7's default, kept for A/B; matches the oracle's exp mode 1.
//   ALPHA_FAST (dgr_set_option("alpha_mode", 1) = "fast_alpha"): the conic pre-scaled by log2(e), power from two fused
//       multiply-adds, one v_exp_f32, v_rcp_f32 -- every operation accurate to an ulp, gradients up to 6e-5 abs away at
//       config 3.
// Measured at config 3 (profiles/r5/alpha_modes_summary.txt, profiles/r8/ab_alpha_modes.txt; end-to-end max |dL_dview - oracle|,
// forward / backward blend kernel): GLIBC 1.8e-7, 128 / 240 us; FAST 5.8e-5, 98-103 / 193-204 us; the reference association
// with a hi/lo-corrected v_exp_f32 1.4e-5, 107 / 219 us; with ocml's expf and the compiler's IEEE division (round 2's
// "exact" build) 6.6e-6, 115 / 271 us.  Only a path that reproduces the restatement's bits brings the alpha image itself to
// the restatement's.
enum { ALPHA_REF = 0, ALPHA_FAST = 1, ALPHA_GLIBC = 2 };
template <int AM>
struct AlphaPath {
    static constexpr bool LOG2 = (AM == ALPHA_FAST);               // staged conic scaled by log2(e): p2 = log2(e) power
    static constexpr float PSCALE = LOG2 ? LOG2E : 1.0f;
    static constexpr float PUNSCALE = LOG2 ? LN2 : 1.0f;
    static constexpr bool TABLE = (AM == ALPHA_GLIBC);             // needs the workgroup's copy of EXP2F_TABLE
};
// o G for one pair: alpha before the 0.99 clamp.  p2 = pair_p2<AM>(); `tab` = LDS copy of EXP2F_TABLE (ALPHA_GLIBC only)
// UNTESTED_ARG: the caller has not bounded p2 from below (the backward blend; the forward's log-domain pre-test has)
template <int AM, bool UNTESTED_ARG = false>
__device__ __forceinline__ float alpha_raw(float o, float p2, const uint64_t* tab) {
    if (AM == ALPHA_FAST) return o * __builtin_amdgcn_exp2f(p2);
    if (AM == ALPHA_GLIBC) return o * exp_glibc<UNTESTED_ARG>(p2, tab);
    return o * exp_p32<UNTESTED_ARG>(p2);
}
// T / om for the backward's transmittance; `inv` ~ 1 / om for the terms that are not amplified
template <int AM>
__device__ __forceinline__ float t_div(float T, float om, float& inv) {
    if (AM == ALPHA_FAST) { inv = __builtin_amdgcn_rcpf(om); return T * inv; }
    return div_ref(T, om, inv);
}

// NB = instances staged per batch (<= 256, one per thread).  The forward uses 256; the light backward 128, which
// halves its LDS footprint (it is occupancy-bound: see DESIGN.md s4.2).
// LIST_T = type of a list entry: unsigned short (the forward: its LDS footprint decides 8 workgroups per CU) or uint32_t (the
// backward kernels: two entries arrive as one 8-byte read, ready to be used as addresses -- parting two 16-bit halves costs a
// v_lshrrev_b32 at 4.2 cycles and a mask at 2.4 per two entries, in loops bound by vector issue).
// NLISTS = 4: one list per quadrant wave; 8: one per HALF of a quadrant (lanes 0-31 = its upper four pixel rows, 32-63 = its
// lower four: build_half_lists) -- the light forward.  KEEP_ID = false: the Gaussian ids are not kept in LDS (the light forward
// re-reads them from the tile list when it writes the tags back).
template <int NB, typename LIST_T = unsigned short, int NLISTS = 4, bool KEEP_ID = true>
struct StagedT {
    typedef LIST_T list_t;
    static constexpr bool HAS_ID = KEEP_ID;
    static constexpr int SLOTS = NB;
    static constexpr int SENTINEL = NB;       // record slot that can never contribute (opacity 0)
    static constexpr int LIST_LD = NB + 8;    // list row: NB entries + sentinel padding, 8-byte aligned rows
    float4 rec[2 * (NB + 1)];  // [2*slot] = {x, y, a2, c2}, [2*slot+1] = {b2, opacity, slot (int bits), lthr}
                               //  p2 = dx*(a2*dx + b2*dy) + c2*dy*dy = log2(e) * power (pair_p2)
    float4 rgbd[NB + 1];       // {r, g, b, depth}; [NB] = zeros (the sentinel's entry: the branch-free backward reads it)
    uint32_t id[KEEP_ID ? NB : 1];
    LIST_T list[NLISTS][LIST_LD];     // per consumer wave (half-wave): byte offsets (slot * 32) into rec, tile-list order
    int cnt4[4][NLISTS];              // [staging wave][consumer] entries contributed
};
using Staged = StagedT<DGR_TILE_PIX>;

template <bool HALF_CODES>
__device__ __forceinline__ unsigned reach_code(const float4& q0, const float4& q1, float l2, float tile_x0, float tile_y0);

// Stage one instance and return the 4-bit "may touch quadrant" code.
// A quadrant is kept when the bounding box of the region alpha >= 15/255, i.e. q(d) <= tau = 2 ln(255 o / 15), reaches
// one of its pixels; the box carries a safety margin far above the rounding of the per-pixel evaluation (and of the
// fast rcp / sqrt used here), so every dropped (pixel, Gaussian) pair is one the per-pixel test rejects.
// (An exact ellipse-vs-quadrant test was measured: it removes almost no list entries beyond the box -- the
// iterations without a valid lane are finished pixels and sub-pixel splats -- and costs more VALU than it saves.)
// HALF_CODES: 8 bits, bit 2 q + h = half h (pixel rows 4 h .. 4 h + 3) of quadrant q.
template <int AM, bool HALF_CODES = false, class S>
__device__ __forceinline__ unsigned stage_one(S& s, int slot, uint32_t gid, const float4* __restrict__ rec,
                                              float tile_x0, float tile_y0) {
    const float4 q0 = rec[DGR_REC_STRIDE * (size_t)gid + 0];
    const float4 q1 = rec[DGR_REC_STRIDE * (size_t)gid + 1];
    const float4 q2 = rec[DGR_REC_STRIDE * (size_t)gid + 2];
    const float o = q0.w;
    // log-domain threshold: alpha >= 15/255 <=> p2 >= log2(15/(255 o)); the loop compares against a slightly lower
    // value and re-tests alpha itself on the rare path, so decisions are those of the linear-domain test.
    const float l2 = __log2f(o * (255.0f / 15.0f));  // = tau / (2 ln 2)
    constexpr float PSCALE = AlphaPath<AM>::PSCALE;
    float lthr = (o > 0.f) ? (-l2 * (AlphaPath<AM>::LOG2 ? 1.0f : LN2) - 1.0e-4f) : 3.0e38f;
    float zword = __int_as_float(slot);
    if (HALF_CODES) {
        // The slot rides in the low 8 bits of the threshold, which is first rounded AWAY from zero to a multiple of 256 ulp: a negative
        // threshold (the only kind anything can pass: p2 <= 0) gets at most 511 ulp more permissive, whatever its magnitude -- a few
        // more pairs reach the exact test, which decides as before --
        // and the record's third word becomes four spare BYTES: byte w = "some pixel of the LOWER half of quadrant wave w blended
        // this instance" (the upper halves' bytes are the kernel's `hit` words) -- contribution tags per half, free of LDS.
        // (an infinite opacity has the threshold -inf, whose bits would become a NaN's with the slot in them -- and nothing passes a
        //  NaN, where the reference blends such a Gaussian at alpha = min(0.99, inf) = 0.99: a finite floor first)
        lthr = fmaxf(lthr, -3.0e38f);
        lthr = __int_as_float(((__float_as_int(lthr) + 0xFF) & ~0xFF) | slot);
        zword = 0.f;
    }
    s.rec[2 * slot] = make_float4(q0.x, q0.y, -0.5f * PSCALE * q1.x, -0.5f * PSCALE * q1.z);
    s.rec[2 * slot + 1] = make_float4(-PSCALE * q1.y, o, zword, lthr);
    s.rgbd[slot] = make_float4(q2.x, q2.y, q2.z, q0.z);
    if (S::HAS_ID) s.id[slot] = gid;
    return reach_code<HALF_CODES>(q0, q1, l2, tile_x0, tile_y0);
}
// which quadrants (HALF_CODES: which halves of the quadrants) the box of the region alpha >= 15/255 reaches; l2 = log2(255 o / 15)
template <bool HALF_CODES>
__device__ __forceinline__ unsigned reach_code(const float4& q0, const float4& q1, float l2, float tile_x0, float tile_y0) {
    const float tau = 2.0f * 0.6931471805599453f * l2;
    const float det = q1.x * q1.z - q1.y * q1.y;
    if (!(tau > 0.0f)) return 0u;                                       // opacity below 15/255: can never contribute
    if (!(det > 0.0f && q1.x > 0.0f && q1.z > 0.0f)) return HALF_CODES ? 0xFFu : 0xFu;  // degenerate conic: do not cull
    const float k = tau * __builtin_amdgcn_rcpf(det);
    const float hx = __builtin_amdgcn_sqrtf(k * q1.z) * 1.001f + 0.05f, hy = __builtin_amdgcn_sqrtf(k * q1.x) * 1.001f + 0.05f;
    const float gx = q0.x - tile_x0, gy = q0.y - tile_y0;              // centre relative to the tile origin
    const bool xl = (gx + hx >= 0.0f) && (gx - hx <= 7.0f), xr = (gx + hx >= 8.0f) && (gx - hx <= 15.0f);
    if (HALF_CODES) {
        const float lo = gy - hy, hi = gy + hy;
        const unsigned xm = (xl ? 0x33u : 0u) | (xr ? 0xCCu : 0u);    // quadrants 0, 2 are the left ones
        const unsigned ym = ((hi >= 0.0f) && (lo <= 3.0f) ? 0x05u : 0u) | ((hi >= 4.0f) && (lo <= 7.0f) ? 0x0Au : 0u) |
                            ((hi >= 8.0f) && (lo <= 11.0f) ? 0x50u : 0u) | ((hi >= 12.0f) && (lo <= 15.0f) ? 0xA0u : 0u);
        return xm & ym;
    }
    const bool yt = (gy + hy >= 0.0f) && (gy - hy <= 7.0f), yb = (gy + hy >= 8.0f) && (gy - hy <= 15.0f);
    return (xl && yt ? 1u : 0u) | (xr && yt ? 2u : 0u) | (xl && yb ? 4u : 0u) | (xr && yb ? 8u : 0u);
}

// Contribution tags.  The forward blend kernel marks every tile-list entry with the set of quadrant waves in which at least one
// pixel blended it.  A (pixel, Gaussian) pair is valid in the backward exactly when the forward blended it (same alpha
// expression, and position <= the pixel's last contributor), so the backward builds its per-wave lists from the tags: no
// bounding-box test, no wave-level pre-test, and instances no pixel blended are never loaded.
// One BYTE per list entry, per HALF of a quadrant (bit 2 w + h: half h -- pixel rows 4 h .. 4 h + 3, lanes 32 h .. -- of quadrant
// wave w), in an instance-major byte array beside the list; the list itself is not written by any blend kernel (rounds 3-8 kept a
// 4-bit tag in the top bits of the point_list entry as well: 6.4 MB of the light forward's HBM writes per 1080p view for a subset of
// the byte's information -- profiles/r9/fwd_traffic.txt; Gaussian ids still stay below 2^28).  A forward that walked quadrant lists
// sets both halves' bits of a quadrant; a backward that walks quadrant lists folds the two bits of each quadrant.
constexpr int TAG_SHIFT = DGR_TAG_SHIFT;
constexpr uint32_t ID_MASK = DGR_ID_MASK;
enum { TAGS_BYTES_QUADRANT = 1, TAGS_BYTES_HALVES = 2 };

// The tag bytes live in the binning buffer's `pair_cov` / `ranks` bytes, which the binning is done with when the
// blend runs; both blend kernels find them from the capacity the binning kernel left in cursor[2] (sched_flag = cursor + 3).
// INVARIANT: the forward blend writes the byte of EVERY entry of every tile list of its frame (zero for entries nobody blended,
// and for the tail of a list whose tile finished early) -- the bytes underneath are the binning's, and nothing else clears them.
// Whoever keeps a view's state between forward and backward must treat ranks / pair_cov as clobbered by the blend (BinningView,
// dgr_common.h).
__device__ __forceinline__ uint8_t* half_tags(const uint32_t* point_list, const uint32_t* sched_flag) {
    return carve_binning(reinterpret_cast<char*>(const_cast<uint32_t*>(point_list)), (size_t)sched_flag[-1]).pair_cov;
}
// bit w of a 4-bit quadrant code -> bit 2 w (the upper half's place in a tag byte), and back (either half set)
__device__ __forceinline__ uint32_t spread4(uint32_t c) { return (c & 1u) | ((c & 2u) << 1) | ((c & 4u) << 2) | ((c & 8u) << 3); }
__device__ __forceinline__ uint32_t fold8(uint32_t t) {
    t = (t | (t >> 1)) & 0x55u;
    return (t & 1u) | ((t >> 1) & 2u) | ((t >> 2) & 4u) | ((t >> 3) & 8u);
}

// four 0/1 bytes -> four bits
__device__ __forceinline__ uint32_t pack4(uint32_t w) { return (w & 1u) | ((w >> 7) & 2u) | ((w >> 14) & 4u) | ((w >> 21) & 8u); }
// The forward's marks of one staged instance -> its tag byte.  A half-wave forward marks byte w of `up` from the lanes 0-31 of
// quadrant wave w and byte w of `lo` from its lanes 32-63 (the light and the full forward keep `up` in an LDS word per slot and
// `lo` in the spare third word of the staged record: stage_one<AM, true>).
__device__ __forceinline__ uint32_t tag_byte(uint32_t up, uint32_t lo) { return spread4(pack4(up)) | (spread4(pack4(lo)) << 1); }

// backward staging: returns the entry's tag (TAGS: where it lives and in which form it is wanted); untagged entries are not loaded.
template <int AM, int TAGS, class S>
__device__ __forceinline__ unsigned stage_tagged(S& s, int slot, uint32_t entry, const float4* __restrict__ rec,
                                                 const uint8_t* __restrict__ tag8) {
    constexpr float PSCALE = AlphaPath<AM>::PSCALE;
    unsigned code = (unsigned)*tag8;
    if (code == 0u) return 0u;
    if (TAGS == TAGS_BYTES_QUADRANT) code = fold8(code);
    const uint32_t gid = entry & ID_MASK;
    const float4 q0 = rec[DGR_REC_STRIDE * (size_t)gid + 0];
    const float4 q1 = rec[DGR_REC_STRIDE * (size_t)gid + 1];
    const float4 q2 = rec[DGR_REC_STRIDE * (size_t)gid + 2];
    s.rec[2 * slot] = make_float4(q0.x, q0.y, -0.5f * PSCALE * q1.x, -0.5f * PSCALE * q1.z);
    // (.z: 4 * slot = the byte offset of the slot's accumulator column; .w: byte offset of its rgbd entry -- the backward
    //  kernels address LDS with both directly, and compare .z with 4 * (slots at or before the pixel's last contributor))
    s.rec[2 * slot + 1] = make_float4(-PSCALE * q1.y, q0.w, __int_as_float(slot * 4), __int_as_float(slot * 16));
    s.rgbd[slot] = make_float4(q2.x, q2.y, q2.z, q0.z);
    s.id[slot] = gid;
    return code;
}

__device__ __forceinline__ int lanes_below(unsigned long long m) {
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}

// Builds the four per-consumer lists from the staging threads' quadrant codes.  Contains two barriers;
// returns the (uniform) length of the calling wave's list, padded to a multiple of 4 with sentinels.
template <class S>
__device__ __forceinline__ int build_lists(S& s, unsigned code, int tid, int wave, int lane) {
    typedef typename S::list_t LT;
    unsigned long long bal[4];
#pragma unroll
    for (int w = 0; w < 4; w++) {
        bal[w] = __ballot((code >> w) & 1u);
        if (lane == 0) s.cnt4[wave][w] = __popcll(bal[w]);
    }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < 4; w++) {
        if ((code >> w) & 1u) {
            int base = 0;
            for (int sw = 0; sw < wave; sw++) base += s.cnt4[sw][w];
            s.list[w][base + lanes_below(bal[w])] = (LT)(tid * 32);
        }
    }
    const int n = __builtin_amdgcn_readfirstlane(s.cnt4[0][wave] + s.cnt4[1][wave] + s.cnt4[2][wave] + s.cnt4[3][wave]);
    __syncthreads();
    if (lane < 4) s.list[wave][n + lane] = (LT)(S::SENTINEL * 32);  // own list, own wave: program order suffices
    return n;
}

// The same for eight lists, one per half of a quadrant (code: stage_one<AM, true>).  Returns the length of the LONGER of the
// calling wave's two lists; the shorter one is padded with sentinels up to it (+ the unroll's padding), so that one loop
// step serves the upper half's next entry on lanes 0-31 and the lower half's on lanes 32-63.
template <class S>
__device__ __forceinline__ int build_half_lists(S& s, unsigned code, int tid, int wave, int lane) {
    typedef typename S::list_t LT;
    unsigned long long bal[8];
#pragma unroll
    for (int l = 0; l < 8; l++) {
        bal[l] = __ballot((code >> l) & 1u);
        if (lane == 0) s.cnt4[wave][l] = __popcll(bal[l]);
    }
    __syncthreads();
    // lane l < 8 adds up list l's counts ONCE -- those of the staging waves in front of this one (the wave's base in the list) and
    // all four (the list's length) -- and every use below takes its list's numbers with one v_readlane: the per-list reads of the
    // count table cost the staging of a big-splat scene (every instance in all eight lists) a tenth of the forward's instructions
    int before = 0, all = 0;
    if (lane < 8) {
#pragma unroll
        for (int sw = 0; sw < 4; sw++) {
            const int c = s.cnt4[sw][lane];
            before += sw < wave ? c : 0;
            all += c;
        }
    }
#pragma unroll
    for (int l = 0; l < 8; l++) {
        const int base = __builtin_amdgcn_readlane(before, l);
        if ((code >> l) & 1u) s.list[l][base + lanes_below(bal[l])] = (LT)(tid * 32);
    }
    const int l0 = 2 * wave;
    const int n0 = __builtin_amdgcn_readlane(all, l0), n1 = __builtin_amdgcn_readlane(all, l0 + 1);
    __syncthreads();
    const int n = max(n0, n1);
    for (int i = n0 + lane; i < n + 4; i += 64) s.list[l0][i] = (LT)(S::SENTINEL * 32);      // own lists, own wave: program order
    for (int i = n1 + lane; i < n + 4; i += 64) s.list[l0 + 1][i] = (LT)(S::SENTINEL * 32);
    return n;
}

// Paired lists (the mapping backward).  A (quadrant, Gaussian) entry reaches 16 of the wave's 64 pixels on average; 28 % of the
// entries stay in the quadrant's upper half (lanes 0-31), 28 % in its lower half, 44 % touch both.  Two NEIGHBOURS of the wave's
// list of which one lives in the upper and the other in the lower half can share a loop step -- no pixel sees both, so every
// pixel still meets its Gaussians in list order -- and the step's reduction stops before the stage that adds the halves
// (wave_reduce.h).  Every entry is still delivered exactly once with twelve lane-atomics (half-wave lists, where an entry of both
// halves is delivered twice, drown in them: DESIGN.md Appendix A).
//   code8 : stage_tagged<AM, TAGS_BYTES_HALVES>'s code (bit 2 w + h: half h of quadrant wave w)
//   phase 1: the four compacted quadrant lists as in build_lists, entry = record offset | type (1 upper, 2 lower, 3 both);
//   phase 2: every wave pairs its own list (of at most 64 entries: one rank per lane; longer lists stay unpaired): ranks
//            (2 m, 2 m + 1) first, then (2 m + 1, 2 m + 2) where neither was taken -- a window of five entries, read through
//            whole-wave DPP shifts, decides; no scan -- and writes the STEP lists in place: list[2 w] = what lanes 0-31 process at
//            step s, list[2 w + 1] = what lanes 32-63 process (the same entry unless the step is a pair).  0.87 steps per entry.
// Returns the number of steps (wave-uniform); split[c] bit b: step 64 c + b is a pair (the two halves hold different entries).
template <class S>
__device__ __forceinline__ int build_paired_lists(S& s, unsigned code8, int tid, int wave, int lane, unsigned long long (&split)[2]) {
    typedef typename S::list_t LT;
    static_assert(sizeof(LT) == 4 && S::SLOTS <= 128, "paired lists: 32-bit entries, at most two ranks per lane");
    unsigned long long bal[4];
#pragma unroll
    for (int w = 0; w < 4; w++) {
        bal[w] = __ballot(((code8 >> (2 * w)) & 3u) != 0u);
        if (lane == 0) s.cnt4[wave][w] = __popcll(bal[w]);
    }
    __syncthreads();
    int before = 0, all = 0;  // lane w < 4: list w's entries from the staging waves in front of this one / from all four (build_half_lists)
    if (lane < 4) {
#pragma unroll
        for (int sw = 0; sw < 4; sw++) {
            const int c = s.cnt4[sw][lane];
            before += sw < wave ? c : 0;
            all += c;
        }
    }
#pragma unroll
    for (int w = 0; w < 4; w++) {
        const unsigned t = (code8 >> (2 * w)) & 3u;
        const int base = __builtin_amdgcn_readlane(before, w);
        if (t != 0u) s.list[2 * w][base + lanes_below(bal[w])] = (LT)(tid * 32) | t;
    }
    const int n = __builtin_amdgcn_readlane(all, wave);
    __syncthreads();
    LT* const LU = s.list[2 * wave];
    LT* const LL = s.list[2 * wave + 1];
    if (n <= 64) {
        // the usual case (a batch of 128 positions leaves ~22 entries per quadrant): one rank per lane, the window of five types
        // through whole-wave DPP shifts (a lane without a neighbour reads 0 = no entry)
        const bool have = lane < n;
        const unsigned e0 = have ? LU[lane] : 0u;
        const int t0 = (int)(e0 & 3u);
        const int tm1 = __builtin_amdgcn_update_dpp(0, t0, 0x138, 0xf, 0xf, true);   // wave_shr:1
        const int tm2 = __builtin_amdgcn_update_dpp(0, tm1, 0x138, 0xf, 0xf, true);
        const int tp1 = __builtin_amdgcn_update_dpp(0, t0, 0x130, 0xf, 0xf, true);   // wave_shl:1
        const int tp2 = __builtin_amdgcn_update_dpp(0, tp1, 0x130, 0xf, 0xf, true);
        const bool odd = (lane & 1) != 0;
        const bool a_self = odd ? (tm1 * t0 == 2) : (t0 * tp1 == 2);
        const bool a_side = odd ? (tp1 * tp2 == 2) : (tm2 * tm1 == 2);
        const bool p2 = !a_self && !a_side && (odd ? (t0 * tp1 == 2) : (tm1 * t0 == 2));
        const bool first = odd ? p2 : a_self, second = odd ? a_self : p2;
        const unsigned long long keep = __ballot(have && !second);
        const int steps = __popcll(keep);
        if (have) {
            const int st = lanes_below(keep) - (second ? 1 : 0);
            const LT o = e0 & ~3u;
            if (first || second) {
                (t0 == 1 ? LU : LL)[st] = o;
            } else {
                LU[st] = o;
                LL[st] = o;
            }
        }
        if (lane < 4) {
            LU[steps + lane] = (LT)(S::SENTINEL * 32);
            LL[steps + lane] = (LT)(S::SENTINEL * 32);
        }
        split[0] = __ballot(lane < steps && LU[lane] != LL[lane]);
        split[1] = 0ull;
        return steps;
    }
    // more than 64 entries in one quadrant's list of a 128-position batch (big splats, which live in both halves anyway): unpaired
    const LT ea = lane < n ? LU[lane] : 0u, eb = lane + 64 < n ? LU[lane + 64] : 0u;
    if (lane < n) { LU[lane] = ea & ~3u; LL[lane] = ea & ~3u; }
    if (lane + 64 < n) { LU[lane + 64] = eb & ~3u; LL[lane + 64] = eb & ~3u; }
    if (lane < 4) {
        LU[n + lane] = (LT)(S::SENTINEL * 32);
        LL[n + lane] = (LT)(S::SENTINEL * 32);
    }
    split[0] = split[1] = 0ull;
    return n;
}

// v_pk_*_f32 operands: gfx950 issues two fp32 operations per lane with one packed instruction
typedef float f2 __attribute__((ext_vector_type(2)));

// power (ALPHA_FAST: log2(e) * power) of one (pixel, Gaussian) pair from a staged record, and the offsets d = centre - pixel.
// Forward and backward must agree on every decision, so both evaluate exactly this sequence.
//   ALPHA_FAST: a packed subtract, a packed multiply, two fused multiply-adds and one multiply;
//   otherwise : the reference's association without contraction (forward.cu:354, backward.cu:561); the staged a2, c2
//               carry the factor -0.5 and b2 the sign, which are exact.
template <int AM>
__device__ __forceinline__ float pair_p2(const float4& q0, const float4& q1, f2 pxy, f2& dxy) {
    const f2 g = {q0.x, q0.y}, ac = {q0.z, q0.w};
    dxy = g - pxy;
    if (AM == ALPHA_FAST) {
        const f2 m = ac * dxy;                                      // a2 dx, c2 dy
        const float t = __builtin_fmaf(q1.x, dxy.y, m.x);           // a2 dx + b2 dy
        return __builtin_fmaf(dxy.x, t, m.y * dxy.y);
    } else {
#pragma clang fp contract(off)
        const f2 m = (ac * dxy) * dxy;                              // (a2 dx) dx, (c2 dy) dy
        const float B = (q1.x * dxy.x) * dxy.y;
        return (m.x + m.y) + B;
    }
}

// two consecutive list entries: one LDS read (4 bytes, parted by a mask and a shift; 8 bytes with 32-bit entries)
// (`wave`: the list's index -- the wave, or a lane's half-wave list with eight lists)
template <class S>
__device__ __forceinline__ void load2(const S& s, int wave, int k, float4 (&q0)[2], float4 (&q1)[2]) {
    typedef typename S::list_t LT;
    unsigned off[2];
    if (sizeof(LT) == 4) {
        const uint2 pk = *reinterpret_cast<const uint2*>(&s.list[wave][k]);
        off[0] = pk.x;
        off[1] = pk.y;
    } else {
        const unsigned pk = *reinterpret_cast<const unsigned*>(&s.list[wave][k]);
        off[0] = pk & 0xffffu;
        off[1] = pk >> 16;
    }
    const char* base = reinterpret_cast<const char*>(s.rec);
#pragma unroll
    for (int u = 0; u < 2; u++) {
        q0[u] = *reinterpret_cast<const float4*>(base + off[u]);
        q1[u] = *reinterpret_cast<const float4*>(base + off[u] + 16);
    }
}


// sentinel record: p2 = 0 but lthr = +big, so it never passes `p2 >= lthr`; opacity 0, so its alpha is 0.
// TAGGED (backward staging, stage_tagged): .w carries the byte offset of the sentinel's all-zero rgbd entry instead.
template <bool TAGGED = false, class S>
__device__ __forceinline__ void write_sentinel(S& s) {
    constexpr int NB = S::SLOTS;
    s.rec[2 * NB] = make_float4(0.f, 0.f, 0.f, 0.f);
    s.rec[2 * NB + 1] = make_float4(0.f, 0.f, __int_as_float(TAGGED ? NB * 4 : NB), TAGGED ? __int_as_float(NB * 16) : 3.0e38f);
    s.rgbd[NB] = make_float4(0.f, 0.f, 0.f, 0.f);
}

constexpr int NACC = DGR_ACC_STRIDE;          // accumulator components carried per staged instance (<= 16)
constexpr int ACC_LD = DGR_TILE_PIX + 1;      // component-major LDS accumulators, +1 pad for the transposed flush

// flush of the per-batch LDS accumulators: 16 consecutive lanes cover one Gaussian's 64-byte accumulator row
template <int NCOMP, int LD = ACC_LD>
__device__ __forceinline__ void flush_acc(const float* lds_acc, const uint32_t* ids, int cnt, float* global_acc, int tid) {
    const int comp = tid & 15;
    if (comp < NCOMP) {
        for (int r = tid >> 4; r < cnt; r += 16) {
            const float v = lds_acc[comp * LD + r];
            if (v != 0.f) atomicAdd(global_acc + (size_t)ids[r] * DGR_ACC_STRIDE + comp, v);
        }
    }
}

}  // namespace
}  // namespace dgr
