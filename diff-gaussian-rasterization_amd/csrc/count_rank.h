// count_rank.h -- first half of duplicateWithKeys (L/cuda_rasterizer/rasterizer_impl.cu:70-111) as a block-level
// device function: every Gaussian of a 256-thread block takes ONE returning atomic per tile instance (or per same-row
// instance pair) on the tile counters and keeps the arrival ranks, stored Gaussian-major at `off0`.
// Used by the stand-alone count_rank kernel (callback path: the binning buffer is sized from num_rendered first) and,
// fused, by preprocess_fwd (presized path: no separate scan over P, no second pass over the rectangles).
#pragma once
#include "dgr_common.h"

namespace dgr {

constexpr int COUNT_STAGE = 4096;  // ranks staged per 256-Gaussian block (16 KB of LDS)

// `r`: tile rectangle of this thread's Gaussian (empty for culled ones), `off0`: index of its first instance,
// `block_base` / `block_total`: first instance and instance count of the whole block (contiguous in the output),
// `stage`: COUNT_STAGE words of LDS.  Contains one __syncthreads() when the block's ranks fit the stage.
__device__ __forceinline__ void count_and_rank(ushort4 r, uint32_t off0, uint32_t block_base, uint32_t block_total,
                                               uint32_t* tile_count, uint32_t* ranks, int grid_x, int capacity,
                                               uint32_t* stage, int tid) {
    const uint32_t w = (uint32_t)(r.z - r.x), h = (uint32_t)(r.w - r.y);
    const uint32_t n = w * h;
    // Ranks of one block are contiguous in the output: stage them in LDS and write them out coalesced (the per-thread
    // 4-byte stores of the direct form are scattered).  Blocks with more instances than the stage holds, and anything
    // past the capacity, take the direct path / are only counted.
    const bool staged = block_total <= (uint32_t)COUNT_STAGE;
    const bool store = off0 + n <= (uint32_t)capacity;  // past the capacity an instance is still counted, so that
                                                        // scan_tiles sees the true total and flags the overflow
    const uint32_t loc = off0 - block_base;
    // Horizontally adjacent tiles (2p, 2p + 1) share one 64-bit counter {count(2p), count(2p + 1) << 32}: an instance
    // pair in the same row takes ONE returning atomic that bumps both halves (the stage is bound by the number of
    // atomics the memory-side unit retires, not by their width).  Per rectangle row: an unpaired tile first if the row
    // starts at an odd tile, then pairs, then an unpaired last tile.
    const uint32_t lead = r.x & 1u;                       // row starts at an odd tile
    const uint32_t rest = (w > lead) ? w - lead : 0u;
    const uint32_t ops_row = (w ? (w >= lead ? lead : 0u) : 0u) + (rest >> 1) + (rest & 1u);
    const uint32_t n_ops = ops_row * h;
    const int pairs_x = (grid_x + 1) >> 1;
    unsigned long long* cnt64 = reinterpret_cast<unsigned long long*>(tile_count);
    auto put = [&](uint32_t kk, uint32_t v) {
        if (staged) stage[loc + kk] = v;
        else if (store) ranks[off0 + kk] = v;
    };
    // four returning atomics in flight per thread
    for (uint32_t q = 0; q < n_ops; q += 4) {
        unsigned long long got[4];
        uint32_t kk0[4];
        int kind[4];  // 0: none, 1: low half only, 2: high half only, 3: both
#pragma unroll
        for (int u = 0; u < 4; u++) {
            kind[u] = 0;
            const uint32_t qq = q + u;
            if (qq < n_ops) {
                const uint32_t yy = qq / ops_row, o = qq - yy * ops_row;
                uint32_t x;  // first tile of this op, relative to r.x
                bool both = false;
                if (lead && o == 0) {
                    x = 0;
                } else {
                    x = lead + 2u * (o - lead);
                    both = x + 1u < w;
                }
                const uint32_t tx = r.x + x;
                unsigned long long* c = cnt64 + ((size_t)(r.y + yy) * pairs_x + (tx >> 1)) * (DGR_COUNT_STRIDE / 2);
                kk0[u] = yy * w + x;
                if (both) {
                    kind[u] = 3;
                    got[u] = atomicAdd(c, 0x0000000100000001ull);
                } else if (tx & 1u) {
                    kind[u] = 2;
                    got[u] = (unsigned long long)atomicAdd(reinterpret_cast<uint32_t*>(c) + 1, 1u) << 32;
                } else {
                    kind[u] = 1;
                    got[u] = atomicAdd(reinterpret_cast<uint32_t*>(c), 1u);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (kind[u] == 3) {
                put(kk0[u], (uint32_t)got[u]);
                put(kk0[u] + 1u, (uint32_t)(got[u] >> 32));
            } else if (kind[u] == 2) {
                put(kk0[u], (uint32_t)(got[u] >> 32));
            } else if (kind[u] == 1) {
                put(kk0[u], (uint32_t)got[u]);
            }
        }
    }
    if (staged) {
        __syncthreads();
        // The bound is tested per element in 64 bits on purpose.  `block_base >= capacity ? 0 : min(block_total,
        // capacity - block_base)` becomes llvm.usub.sat(capacity, block_base), and hipcc 7.2 selects a wave-uniform
        // usub.sat as a plain s_sub_i32 without the saturation (the VALU form gets `clamp` and is right): a block whose
        // run starts past the capacity then wrote all of its ranks behind the buffer (seen as a memory fault by
        // tests/tools/soak_parity.py; tests/test_hip_edge_cases.py::test_overflow_writes_stay_inside_the_state_buffers).
        const unsigned long long cap64 = (unsigned long long)(uint32_t)capacity;
        for (uint32_t i = tid; i < block_total; i += 256)
            if ((unsigned long long)block_base + i < cap64) ranks[block_base + i] = stage[i];
    }
}

// In-block exclusive scan of the instance counts: returns this thread's offset inside the block, *total = block total.
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t n, uint32_t* wtot /* 4 words of LDS */, int tid,
                                                         uint32_t* total) {
    const int lane = tid & 63, wave = tid >> 6;
    uint32_t incl = n;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t v = __shfl_up(incl, off, 64);
        if (lane >= off) incl += v;
    }
    if (lane == 63) wtot[wave] = incl;
    __syncthreads();
    uint32_t before = 0;
    for (int ww = 0; ww < wave; ww++) before += wtot[ww];
    *total = wtot[0] + wtot[1] + wtot[2] + wtot[3];
    return before + incl - n;
}

}  // namespace dgr
