// tile_sort.h -- compare-exchange networks on 64-bit keys (depth bits << 32 | gaussian id) shared by the binning kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

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
template <int MASK, int LOWBIT>
__device__ __forceinline__ uint32_t cmpx32(uint32_t v, int lane) {
    const uint32_t o = xor_lane32<MASK>(v);
    const bool lower = (lane & LOWBIT) == 0;
    return lower ? min(v, o) : max(v, o);
}
template <int D>
__device__ __forceinline__ uint32_t disperse32(uint32_t v, int lane) {
    if constexpr (D > 0) return disperse32<D / 2>(cmpx32<D, D>(v, lane), lane);
    else return v;
}
template <int SIZE>
__device__ __forceinline__ uint32_t merge_stage32(uint32_t v, int lane) {
    return disperse32<SIZE / 4>(cmpx32<SIZE - 1, SIZE / 2>(v, lane), lane);
}
__device__ __forceinline__ uint32_t chunk_sort64_32(uint32_t v, int lane) {
    v = merge_stage32<2>(v, lane);
    v = merge_stage32<4>(v, lane);
    v = merge_stage32<8>(v, lane);
    v = merge_stage32<16>(v, lane);
    v = merge_stage32<32>(v, lane);
    return merge_stage32<64>(v, lane);
}
constexpr uint32_t TRUNC_SLOT_BITS = 10, TRUNC_SLOT_MASK = (1u << TRUNC_SLOT_BITS) - 1u;

// `list`: the n <= 64 NCH <= 1024 keys of one tile in LDS (overwritten).  IDS_OUT: dst_ids[0 .. n) receives the ids in key
// order; otherwise the list itself ends up sorted (dst_ids unused).  One wave; no barrier.
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
template <int NCH, bool IDS_OUT = true>
__device__ __forceinline__ void sort_wave_trunc(uint64_t* list, int n, uint32_t* __restrict__ dst_ids, int lane) {
    static_assert(NCH * 64 <= (1 << TRUNC_SLOT_BITS), "slot bits");
    uint32_t v[NCH];
#pragma unroll
    for (int c = 0; c < NCH; c++) {
        const int e = c * 64 + lane;
        v[c] = (e < n) ? (((uint32_t)(list[e] >> 32) & ~TRUNC_SLOT_MASK) | (uint32_t)e) : 0xffffffffu;
        if ((c & 1) && NCH > 1) v[c] = ~v[c];
        v[c] = chunk_sort64_32(v[c], lane);
    }
#pragma unroll
    for (int kc = 2; kc <= NCH; kc <<= 1) {  // merge blocks of kc chunks
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            const bool have = ((c / (kc / 2)) & 1) != 0;
            const bool want = (kc < NCH) && (((c / kc) & 1) != 0);
            if (have != want) v[c] = ~v[c];
        }
#pragma unroll
        for (int jc = kc / 2; jc >= 1; jc >>= 1) {
#pragma unroll
            for (int c = 0; c < NCH; c++)
                if ((c & jc) == 0) {
                    const uint32_t lo = min(v[c], v[c + jc]), hi = max(v[c], v[c + jc]);
                    v[c] = lo; v[c + jc] = hi;
                }
        }
#pragma unroll
        for (int c = 0; c < NCH; c++) v[c] = disperse32<32>(v[c], lane);
    }
    // full keys in (truncated depth, slot) order: every lane fetches its entries' keys, THEN the list is overwritten
    uint64_t key[NCH];
#pragma unroll
    for (int c = 0; c < NCH; c++) key[c] = (c * 64 + lane < n) ? list[v[c] & TRUNC_SLOT_MASK] : KEY_INF;
    wave_lds_fence();
#pragma unroll
    for (int c = 0; c < NCH; c++)
        if (c * 64 + lane < n) list[c * 64 + lane] = key[c];
    wave_lds_fence();
    // fix-up of the runs with equal upper depth bits
    int final_pos[NCH];
#pragma unroll
    for (int c = 0; c < NCH; c++) {
        const int i = c * 64 + lane;
        final_pos[c] = i;
        if (i < n) {
            const uint32_t t = (uint32_t)(key[c] >> 32) >> TRUNC_SLOT_BITS;
            int pos = i;
            for (int j = i - 1; j >= 0; j--) {           // run-mates in front of i: those with a LARGER key move behind it
                const uint64_t o = list[j];
                if (((uint32_t)(o >> 32) >> TRUNC_SLOT_BITS) != t) break;
                if (o > key[c]) pos--;
            }
            for (int j = i + 1; j < n; j++) {            // run-mates behind i: those with a SMALLER key move in front of it
                const uint64_t o = list[j];
                if (((uint32_t)(o >> 32) >> TRUNC_SLOT_BITS) != t) break;
                if (o < key[c]) pos++;
            }
            if (IDS_OUT) dst_ids[pos] = (uint32_t)key[c];
            final_pos[c] = pos;
        }
    }
    if (!IDS_OUT) {  // every lane has finished reading its neighbours: the keys go to their final positions
        wave_lds_fence();
#pragma unroll
        for (int c = 0; c < NCH; c++)
            if (c * 64 + lane < n) list[final_pos[c]] = key[c];
        wave_lds_fence();
    }
}

// 512 < n <= 1024: four parts of 256 sorted as above (keys in place), then merged by rank -- an entry's final position is
// its position in its own part plus the number of entries of every OTHER part in front of it (binary searches in LDS; keys
// are unique).  A single 1024-entry pass would hold 48 registers per lane around its key fetch, and two 512-entry passes
// still spill at the 80 registers bin_tiles has.
__device__ __forceinline__ void sort_wave_trunc_1024(uint64_t* list, int n, uint32_t* __restrict__ dst_ids, int lane) {
    constexpr int Q = 256;
    const int parts = (n + Q - 1) / Q;
    for (int p = 0; p < parts; p++) sort_wave_trunc<4, false>(list + p * Q, min(Q, n - p * Q), nullptr, lane);
    for (int i = lane; i < n; i += 64) {
        const uint64_t k = list[i];
        const int mine = i / Q;
        int pos = i - mine * Q;
        for (int p = 0; p < parts; p++) {
            if (p == mine) continue;
            const uint64_t* other = list + p * Q;
            int lo = 0, hi = min(Q, n - p * Q);  // lower bound of k in part p
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (other[mid] < k) lo = mid + 1; else hi = mid;
            }
            pos += lo;
        }
        dst_ids[pos] = (uint32_t)k;
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

// ---- the full 64-bit keys in registers: lists of 513 .. 1024 entries where they are the rule (bin_tiles<LONG_LISTS>) --------------
// (the 32-bit form would hold 16 words + 16 keys per lane around its key fetch, and the four-part merge above is a chain of
//  dependent LDS reads: 159 against 116 us at config 4's 810 entries per tile)
__device__ __forceinline__ void cmpx_regs(uint64_t& a, uint64_t& b) {  // a <- min, b <- max
    const bool sw = b < a;
    const uint64_t lo = sw ? b : a, hi = sw ? a : b;
    a = lo; b = hi;
}
template <int NCH>
__device__ __forceinline__ void sort_wave_regs(const uint64_t* __restrict__ src, int n, uint32_t* __restrict__ dst_ids, int lane) {
    uint64_t v[NCH];
#pragma unroll
    for (int c = 0; c < NCH; c++) {
        const int e = c * 64 + lane;
        v[c] = (e < n) ? src[e] : KEY_INF;
        if ((c & 1) && NCH > 1) v[c] = ~v[c];
        v[c] = chunk_sort64(v[c], lane);
    }
#pragma unroll
    for (int kc = 2; kc <= NCH; kc <<= 1) {  // merge blocks of kc chunks
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            const bool have = ((c / (kc / 2)) & 1) != 0;                 // complemented by the previous stage
            const bool want = (kc < NCH) && (((c / kc) & 1) != 0);       // this block must come out descending
            if (have != want) v[c] = ~v[c];
        }
#pragma unroll
        for (int jc = kc / 2; jc >= 1; jc >>= 1) {
#pragma unroll
            for (int c = 0; c < NCH; c++)
                if ((c & jc) == 0) cmpx_regs(v[c], v[c + jc]);
        }
#pragma unroll
        for (int c = 0; c < NCH; c++) v[c] = chunk_tail64(v[c], lane);
    }
#pragma unroll
    for (int c = 0; c < NCH; c++) {
        const int e = c * 64 + lane;
        if (e < n) dst_ids[e] = (uint32_t)v[c];
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
