// tile_sort.h -- compare-exchange networks on 64-bit keys (depth bits << 32 | gaussian id) shared by the binning kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "tile_sort_net.h"

namespace dgr {
namespace {

// ---- per-tile sort ------------------------------------------------------------------------------
// All-ascending bitonic network ("flip" then "disperse" steps): every comparator moves the smaller key to the
// lower index, so +inf padding never moves and the network sorts any n <= np2.
// gfx950 shape: a wave sorts a 64-key chunk entirely in registers -- one key per lane, partner exchange with
// __shfl_xor (ds_bpermute: the LDS crossbar, no bank conflicts, no barrier): all 21 steps of sizes 2..64.  Larger
// merge stages do their cross-wave steps (distance >= 64) on the LDS array with a barrier each, then return to
// registers for distances 32..1.  n = 256 costs 7 barriers instead of the 36 of a plain LDS network.
// Partner exchange lane ^ MASK.  Inside a 16-lane row the DPP network does it on the vector pipe (quad_perm for 1, 2, 3;
// row_half_mirror = ^7, row_mirror = ^15; ^4 = ^7 then ^3, ^8 = ^15 then ^7): 26 of the 33 compare-exchange steps of a 256-key
// tile.  ds_bpermute -- the LDS crossbar, which the four SIMDs of a CU share -- is left with the 7 steps that cross rows (16, 31,
// 32, 63).  Measured in round 2 (every exchange through ds_bpermute): the register steps were 20 of the kernel's 30 us and
// bound by that crossbar.
template <int CTRL>
__device__ __forceinline__ unsigned dpp_mov(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
}
template <int MASK>
__device__ __forceinline__ unsigned xor_lane32(unsigned v) {
    if constexpr (MASK == 1) return dpp_mov<0xB1>(v);        // quad_perm [1,0,3,2]
    else if constexpr (MASK == 2) return dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
    else if constexpr (MASK == 3) return dpp_mov<0x1B>(v);   // quad_perm [3,2,1,0]
    else if constexpr (MASK == 7) return dpp_mov<0x141>(v);  // row_half_mirror
    else if constexpr (MASK == 15) return dpp_mov<0x140>(v); // row_mirror
    else if constexpr (MASK == 4) return dpp_mov<0x1B>(dpp_mov<0x141>(v));
    else if constexpr (MASK == 8) return dpp_mov<0x141>(dpp_mov<0x140>(v));
    else return (unsigned)__shfl_xor(v, MASK, 64);
}
template <int MASK, int LOWBIT>
__device__ __forceinline__ uint64_t cmpx(uint64_t v, int lane) {
    const uint64_t o = ((uint64_t)xor_lane32<MASK>((unsigned)(v >> 32)) << 32) | xor_lane32<MASK>((unsigned)v);
    const bool lower = (lane & LOWBIT) == 0;  // this lane holds the lower index of the pair
    return (lower == (v < o)) ? v : o;         // lower keeps the minimum, upper the maximum
}
template <int D>
__device__ __forceinline__ uint64_t disperse(uint64_t v, int lane) {  // disperse steps D, D/2, .., 1
    if constexpr (D > 0) return disperse<D / 2>(cmpx<D, D>(v, lane), lane);
    else return v;
}
template <int SIZE>
__device__ __forceinline__ uint64_t merge_stage(uint64_t v, int lane) {
    // flip: partner = lane ^ (SIZE-1); the lower half has bit SIZE/2 clear
    return disperse<SIZE / 4>(cmpx<SIZE - 1, SIZE / 2>(v, lane), lane);
}
__device__ __forceinline__ uint64_t chunk_sort64(uint64_t v, int lane) {
    v = merge_stage<2>(v, lane);
    v = merge_stage<4>(v, lane);
    v = merge_stage<8>(v, lane);
    v = merge_stage<16>(v, lane);
    v = merge_stage<32>(v, lane);
    return merge_stage<64>(v, lane);
}
__device__ __forceinline__ uint64_t chunk_tail64(uint64_t v, int lane) { return disperse<32>(v, lane); }  // disperse steps 32..1

constexpr uint64_t KEY_INF = ~0ull;

// ---- a whole tile list in ONE wave's registers ---------------------------------------------------------------------------
// n <= 64 NCH entries, element i = 64 c + lane in v[c].  Chunks are sorted in registers (the DPP network above); the stages
// above 64 pair element i with i ^ j for j >= 64, i.e. chunk c with chunk c ^ (j / 64) AT THE SAME LANE: plain register
// compare-exchanges, no cross-lane traffic, no LDS, no barrier.  That needs the standard bitonic network (blocks sorted in
// alternating directions) rather than the all-ascending flip network of the chunk sort -- obtained by keeping the chunks of
// a block that must come out descending COMPLEMENTED (~word: ascending in the complemented domain = descending in the true
// one), so every compare-exchange is the ascending one.
//
// A 64-bit compare-exchange costs two cross-lane moves, a 64-bit compare and two selects per step (~8 instructions with the
// lane-role logic); the list is at most 1024 entries long, so its order is carried by ONE 32-bit word per entry instead:
// (depth bits with the low 10 bits cleared) | slot, slot = the entry's position in the unsorted LDS list.  The network then
// needs a move, v_min_u32, v_max_u32 and a select per step, and the cross-chunk steps are a min and a max.  What the word
// cannot order -- entries whose depths agree in the upper 22 bits (2^-13 relative: a handful per tile) -- is settled by a
// short fix-up on the full keys: the entries of such a run are adjacent after the sort, and each takes the run's start plus
// the number of its run-mates with a smaller full key as its final position.
constexpr uint32_t TRUNC_SLOT_BITS = 10, TRUNC_SLOT_MASK = (1u << TRUNC_SLOT_BITS) - 1u;

__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
// minimum and maximum of a word over the wave, in every lane (as scalars): four DPP steps inside the rows, then the four row
// results through v_readlane
__device__ __forceinline__ void wave_min_max_u32(uint32_t& mn, uint32_t& mx) {
    mn = min(mn, xor_lane32<1>(mn));  mx = max(mx, xor_lane32<1>(mx));
    mn = min(mn, xor_lane32<2>(mn));  mx = max(mx, xor_lane32<2>(mx));
    mn = min(mn, xor_lane32<7>(mn));  mx = max(mx, xor_lane32<7>(mx));
    mn = min(mn, xor_lane32<15>(mn)); mx = max(mx, xor_lane32<15>(mx));
    const uint32_t a0 = __builtin_amdgcn_readlane(mn, 0), a1 = __builtin_amdgcn_readlane(mn, 16), a2 = __builtin_amdgcn_readlane(mn, 32),
                   a3 = __builtin_amdgcn_readlane(mn, 48);
    const uint32_t b0 = __builtin_amdgcn_readlane(mx, 0), b1 = __builtin_amdgcn_readlane(mx, 16), b2 = __builtin_amdgcn_readlane(mx, 32),
                   b3 = __builtin_amdgcn_readlane(mx, 48);
    mn = min(min(a0, a1), min(a2, a3));
    mx = max(max(b0, b1), max(b2, b3));
}

// ---- a whole tile list in ONE wave's registers ---------------------------------------------------------------------------
// `list`: the n <= 64 NCH <= 1024 keys of one tile in LDS (overwritten).  IDS_OUT: dst_ids[0 .. n) receives the ids in key
// order; otherwise the list itself ends up sorted (dst_ids unused).  One wave; no barrier.
//
// A 64-bit compare-exchange costs two cross-lane moves, a 64-bit compare and two selects; the list is at most 1024 entries long,
// so its order is carried by ONE 32-bit word per entry instead: 22 bits of depth | the entry's position in the unsorted LDS list
// (10 bits).  Round 9: the 22 bits are not the depth's upper bits but the upper bits of (depth bits - the list's smallest) shifted
// left by the leading zeros of the list's depth RANGE -- a tile's depths span a few octaves at most (synth-v1: 2.6, 7 bits
// gained; a wall seen by a SLAM camera: a fraction of one), so two entries share a word prefix about once in a hundred
// tiles instead of in six tiles of ten, and the runs real maps produce (surfaces: many Gaussians within 2^-13 of each other's
// depth) stay short.  What the word cannot order -- entries that agree in those 22 bits -- is settled by a fix-up on the full
// keys: the entries of such a run are adjacent after the sort, and each takes the run's start plus the number of its run-mates
// with a smaller full key as its final position.
//
// The network itself is generated (gen_tile_sort_net.py -> tile_sort_net.h; word index = lane * NCH + register: the frequent
// small distances are register-to-register, v_med3_u32 and bank-masked DPP min / max replace the compare-select pairs,
// every step is issued for all of a lane's registers before the next).  Round 8's routine (profiles/microbench/tile_sort_r8.h)
// needed 3.9 us for 200 entries and 46 us for 1000 (lists above 512 entries went through a four-part rank merge of dependent
// LDS reads because the three-registers-per-entry epilogue spilled from NCH = 8 on at bin_tiles' 80 registers):
// profiles/r9/wave_sort.txt.
template <int NCH, bool IDS_OUT = true>
__device__ __forceinline__ void sort_wave_trunc(uint64_t* list, int n, uint32_t* __restrict__ dst_ids, int lane) {
    static_assert(NCH * 64 <= (1 << TRUNC_SLOT_BITS), "slot bits");
    // (LDS accesses below are unconditional with clamped indices wherever a stray read is harmless: a condition per access costs
    //  an exec-mask region, a branch and a wait each -- 3 of the first version's 7 us at 1000 entries)
    uint32_t v[NCH];
    uint2* const list2 = reinterpret_cast<uint2*>(list);  // .x = Gaussian id, .y = depth bits
    const int last = n - 1;
    uint32_t mn = 0xffffffffu, mx = 0u;
#pragma unroll
    for (int c = 0; c < NCH; c++) {  // (where a word enters the network does not matter: coalesced reads)
        const int e = c * 64 + lane;
        v[c] = list2[min(e, last)].y;
        mn = min(mn, e < n ? v[c] : 0xffffffffu);
        mx = max(mx, e < n ? v[c] : 0u);
    }
    wave_min_max_u32(mn, mx);
    const uint32_t shift = (uint32_t)__builtin_clz((mx - mn) | 1u);
#pragma unroll
    for (int c = 0; c < NCH; c++) {
        const int e = c * 64 + lane;
        v[c] = (e < n) ? ((((v[c] - mn) << shift) & ~TRUNC_SLOT_MASK) | (uint32_t)e) : 0xffffffffu;
    }
    const NetLane k = net_lane(lane);
    sort_net<NCH>(v, k);
    // the word of sorted position p = lane * NCH + c is in v[c].  Two neighbours with the same prefix differ in their slot bits only
    // (a padding word, all ones, can follow an entry whose prefix is all ones: hence the position tests)
    bool tie = false;
#pragma unroll
    for (int c = 0; c + 1 < NCH; c++) tie |= ((v[c] ^ v[c + 1]) <= TRUNC_SLOT_MASK) && (lane * NCH + c + 1 < n);
    {
        const uint32_t last_of_left = (uint32_t)__shfl_up((int)v[NCH - 1], 1, 64);
        tie |= lane > 0 && ((last_of_left ^ v[0]) <= TRUNC_SLOT_MASK) && (lane * NCH < n);
    }
    const bool any_tie = __ballot(tie) != 0ull;
    if constexpr (IDS_OUT) {
        if (!any_tie) {
            // ---- the rule (no two entries of the list share a prefix): the ids alone, transposed through LDS so that the global
            // stores are coalesced -- every lane fetches its entries' ids, THEN the list's bytes are overwritten (16-byte rows of
            // four ids per lane and register group; a lane with an entry writes all its rows: the list's 8 n bytes hold the
            // 4 (n + NCH - 1) + 8 bytes that takes, n being above 32 NCH)
            uint32_t id[NCH];
#pragma unroll
            for (int c = 0; c < NCH; c++) id[c] = list2[min((int)(v[c] & TRUNC_SLOT_MASK), last)].x;
            if constexpr (NCH == 1) {
                if (lane < n) dst_ids[lane] = id[0];
                return;
            } else {
                wave_lds_fence();
                uint32_t* const tr = reinterpret_cast<uint32_t*>((reinterpret_cast<uintptr_t>(list) + 15u) & ~(uintptr_t)15u);
                if (lane * NCH < n) {
                    if constexpr (NCH == 2) {
                        *reinterpret_cast<uint2*>(tr + lane * 2) = make_uint2(id[0], id[1]);
                    } else {
#pragma unroll
                        for (int g = 0; g < NCH / 4; g++)
                            *reinterpret_cast<uint4*>(tr + lane * NCH + 4 * g) = make_uint4(id[4 * g], id[4 * g + 1], id[4 * g + 2], id[4 * g + 3]);
                    }
                }
                wave_lds_fence();
#pragma unroll
                for (int c = 0; c < NCH; c++) id[c] = tr[min(c * 64 + lane, last)];
#pragma unroll
                for (int c = 0; c < NCH; c++)
                    if (c * 64 + lane < n) dst_ids[c * 64 + lane] = id[c];
                return;
            }
        }
    }
    // ---- some entries share a prefix (or the caller wants the KEYS sorted in place): every lane fetches its entries' keys, THEN
    // the list is overwritten in word order
    uint32_t id[NCH];
#pragma unroll
    for (int c = 0; c < NCH; c++) {
        const uint2 q = list2[min((int)(v[c] & TRUNC_SLOT_MASK), last)];
        id[c] = q.x;
        v[c] = q.y;
    }
    wave_lds_fence();
#pragma unroll
    for (int c = 0; c < NCH; c++)
        if (lane * NCH + c < n) list2[lane * NCH + c] = make_uint2(id[c], v[c]);
    wave_lds_fence();
    if constexpr (!IDS_OUT) {
        if (!any_tie) return;  // word order IS key order
    }
    // from here on position i = 64 c + lane (coalesced).  An entry that shares its prefix with a neighbour takes its run's start
    // plus the number of its run-mates with a smaller full key as its final position (the run is adjacent after the sort).
    auto prefix = [&](uint32_t depth_bits) { return ((depth_bits - mn) << shift) >> TRUNC_SLOT_BITS; };
    auto final_position = [&](int i, uint2 q) -> int {
        const uint32_t t = prefix(q.y);
        const uint64_t key = ((uint64_t)q.y << 32) | q.x;
        int pos = i;
        for (int j = i - 1; j >= 0; j--) {           // run-mates in front of i: those with a LARGER key move behind it
            const uint64_t o = list[j];
            if (prefix((uint32_t)(o >> 32)) != t) break;
            if (o > key) pos--;
        }
        for (int j = i + 1; j < n; j++) {            // run-mates behind i: those with a SMALLER key move in front of it
            const uint64_t o = list[j];
            if (prefix((uint32_t)(o >> 32)) != t) break;
            if (o < key) pos++;
        }
        return pos;
    };
    if constexpr (IDS_OUT) {
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            const int i = c * 64 + lane;
            if (i < n) {
                const uint2 q = list2[i];
                dst_ids[final_position(i, q)] = q.x;
            }
        }
    } else {
        uint32_t moved[NCH];
        uint2 kk[NCH];
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            const int i = c * 64 + lane;
            moved[c] = (uint32_t)i;
            kk[c] = make_uint2(0u, 0u);
            if (i < n) {
                kk[c] = list2[i];
                moved[c] = (uint32_t)final_position(i, kk[c]);
            }
        }
        wave_lds_fence();  // every lane has finished reading its neighbours: the moved keys go to their final positions
#pragma unroll
        for (int c = 0; c < NCH; c++)
            if (moved[c] != (uint32_t)(c * 64 + lane)) list2[moved[c]] = kk[c];
        wave_lds_fence();
    }
}

// Lists above the one-wave limit, whole workgroup of NT threads, keys in LDS (no padding needed): parts of PART_Q entries are
// sorted in place by one wave each (sort_list_part; the caller deals the parts of ALL its long lists to its waves at once),
// then, behind a barrier, every list is merged by rank with every thread searching (merge_list_parts).  1135 entries: ~7 us
// where a bitonic workgroup network (padded to 2048: 66 barrier steps) took ~25.
constexpr int PART_Q = 512;
__device__ __forceinline__ int list_parts(int n) { return (n + PART_Q - 1) / PART_Q; }
__device__ __forceinline__ void sort_list_part(uint64_t* list, int n, int p, int lane) {
    sort_wave_trunc<PART_Q / 64, false>(list + p * PART_Q, min(PART_Q, n - p * PART_Q), nullptr, lane);
}
template <int NT>
__device__ __forceinline__ void merge_list_parts(const uint64_t* list, int n, uint32_t* __restrict__ dst_ids, int tid) {
    const int parts = list_parts(n);
    for (int i = tid; i < n; i += NT) {
        const uint64_t k = list[i];
        const int mine = i / PART_Q;
        int pos = i - mine * PART_Q;
        for (int p = 0; p < parts; p++) {
            if (p == mine) continue;
            const uint64_t* other = list + p * PART_Q;
            int lo = 0, hi = min(PART_Q, n - p * PART_Q);  // lower bound of k in part p (keys are unique)
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (other[mid] < k) lo = mid + 1; else hi = mid;
            }
            pos += lo;
        }
        dst_ids[pos] = (uint32_t)k;
    }
}

// ---- a tile list sorted by a whole workgroup of NT threads -----------------------------------------------------------------
// in place in global memory (any n), one barrier per step
template <int NT>
__device__ __forceinline__ void wg_sort_global(uint64_t* gk, int n, int tid) {
    int np2 = 1;
    while (np2 < n) np2 <<= 1;
    const int half = np2 >> 1;
    for (int size = 2; size <= np2; size <<= 1) {
        const int hs = size >> 1;
        for (int t = tid; t < half; t += NT) {
            const int blk = t / hs, off = t - blk * hs;
            const int i = blk * size + off, j = blk * size + (size - 1 - off);
            if (j < n) { const uint64_t x = gk[i], y = gk[j]; if (x > y) { gk[i] = y; gk[j] = x; } }
        }
        __syncthreads();
        for (int d = size >> 2; d > 0; d >>= 1) {
            for (int t = tid; t < half; t += NT) {
                const int blk = t / d, off = t - blk * d;
                const int i = blk * 2 * d + off, j = i + d;
                if (j < n) { const uint64_t x = gk[i], y = gk[j]; if (x > y) { gk[i] = y; gk[j] = x; } }
            }
            __syncthreads();
        }
    }
}

}  // namespace
}  // namespace dgr
