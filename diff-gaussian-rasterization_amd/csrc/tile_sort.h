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
// n <= 64 NCH keys, element i = 64 c + lane in v[c].  Chunks are sorted in registers (chunk_sort64); the stages above 64
// pair element i with i ^ j for j >= 64, i.e. chunk c with chunk c ^ (j / 64) AT THE SAME LANE: plain register
// compare-exchanges, no cross-lane traffic, no LDS, no barrier.  That needs the standard bitonic network (blocks sorted in
// alternating directions) rather than the all-ascending flip network of chunk_sort64 -- obtained here by keeping the
// chunks of a block that must come out descending COMPLEMENTED (~key: ascending in the complemented domain = descending
// in the true one), so every compare-exchange is the ascending one.
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
// in LDS (n <= the array's size; np2 = n rounded up to a power of two >= 64 must fit too), all-ascending network
template <int NT>
__device__ __forceinline__ void wg_sort_lds(uint64_t* sk, int n, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    int np2 = 64;
    while (np2 < n) np2 <<= 1;
    const int chunks = np2 >> 6;
    for (int e = n + tid; e < np2; e += NT) sk[e] = KEY_INF;
    __syncthreads();
    for (int c = wave; c < chunks; c += NT / 64) {
        const int e = c * 64 + lane;
        sk[e] = chunk_sort64(sk[e], lane);
    }
    __syncthreads();
    const int half = np2 >> 1;
    for (int size = 128; size <= np2; size <<= 1) {
        {   // flip across the `size` block: pairs i <-> blk * size + size - 1 - off span all distances (LDS array)
            const int hs = size >> 1;
            for (int t = tid; t < half; t += NT) {
                const int blk = t / hs, off = t - blk * hs;
                const int i = blk * size + off, j = blk * size + (size - 1 - off);
                const uint64_t x = sk[i], y = sk[j];
                if (x > y) { sk[i] = y; sk[j] = x; }
            }
            __syncthreads();
        }
        for (int d = size >> 2; d >= 64; d >>= 1) {
            for (int t = tid; t < half; t += NT) {
                const int blk = t / d, off = t - blk * d;
                const int i = blk * 2 * d + off, j = i + d;
                const uint64_t x = sk[i], y = sk[j];
                if (x > y) { sk[i] = y; sk[j] = x; }
            }
            __syncthreads();
        }
        for (int c = wave; c < chunks; c += NT / 64) {
            const int e = c * 64 + lane;
            sk[e] = chunk_tail64(sk[e], lane);
        }
        __syncthreads();
    }
}
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
