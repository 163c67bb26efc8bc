// binning.hip -- instance binning for gfx950: tile histogram -> ranges, per-tile key emission,
// per-tile sort.
//
// Replaces cub::DeviceScan::InclusiveSum over P, duplicateWithKeys, cub::DeviceRadixSort::SortPairs
// on 64-bit (tile | depth) keys and identifyTileRanges (L/cuda_rasterizer/rasterizer_impl.cu:70-138,
// 283-323).  The reference sorts all R instances globally on (tile id, depth bits) with a STABLE
// radix sort whose input is in ascending Gaussian order, so its result is the unique ascending
// order on the triple (tile id, depth bits, gaussian id).  Here the tile id never enters a key:
//   1. count_rank histograms instances per tile with ONE returning atomic per instance and keeps the returned
//      arrival rank, stored Gaussian-major at the Gaussian's instance offset (block base + in-block scan of
//      tiles_touched: the reference's point_offsets without a device-wide scan kernel);
//   2. scan_tiles turns the histogram into the range table (this IS identifyTileRanges' output);
//   3. emit_instances writes (depth bits << 32 | gaussian id) to slot range.start + rank: a plain scatter.
//      Instance order inside a tile segment is arbitrary, which is irrelevant because
//   4. sort_tiles sorts each segment in LDS -- a total order on unique keys, hence the same point_list as the
//      reference bit for bit.
// Atomics are the scarce resource here: MI355X retires ~26 G global atomic operations/s regardless of scope,
// address spread or whether a value is returned (profiles/microbench/atomics.hip), i.e. ~63 us per 1.65 M; an
// earlier version paid that twice (histogram in preprocess, slot allocation in emit).
// HBM traffic: 4 B + 8 B written, 4 B + 8 B read, 12 B written per instance, against ~6 radix passes x 24 B in
// the reference.
#include "dgr_common.h"
#include "kernels.h"
#include "count_rank.h"
#include <mutex>

namespace dgr {
namespace {

constexpr int SCAN_THREADS = 1024;
constexpr int SORT_THREADS = 256;
constexpr int SORT_LDS_MAX = 2048;  // keys per tile sorted in LDS (16 KB: 8 workgroups per CU; with 4096 keys = 32 KB only
                                    // 5 fit and the latency-bound sort took 37 us instead of 30); larger tiles sort in global memory

__global__ void __launch_bounds__(SCAN_THREADS) scan_tiles_kernel(ImageView img, int tiles, int grid_x, int capacity, int fused) {
    const int pairs_x = (grid_x + 1) >> 1;
    // tile i = (ty, tx): half tx & 1 of the 64-bit pair counter (ty, tx / 2), one pair per cache line (count_rank)
    auto count_of = [&](int i) {
        const int ty = i / grid_x, tx = i - ty * grid_x;
        return img.tile_count[((size_t)ty * pairs_x + (tx >> 1)) * DGR_COUNT_STRIDE + (tx & 1)];
    };
    __shared__ uint32_t wsum[SCAN_THREADS / 64];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int per = (tiles + SCAN_THREADS - 1) / SCAN_THREADS;
    const int lo = min(t * per, tiles), hi = min(lo + per, tiles);
    // Up to 8 counters per thread stay in registers (every frame up to 8192 tiles, i.e. 1080p); larger grids read
    // the counters a second time.  The padded counters are one cache line each, so the loads are issued together.
    constexpr int REG = 8;
    uint32_t c[REG];
    uint32_t s = 0;
    if (per <= REG) {
#pragma unroll
        for (int k = 0; k < REG; k++) {
            c[k] = (lo + k < hi) ? count_of(lo + k) : 0u;
        }
#pragma unroll
        for (int k = 0; k < REG; k++) s += c[k];
    } else {
        for (int i = lo; i < hi; i++) s += count_of(i);
    }
    // wave-level inclusive scan, then the 16 wave totals through LDS: one barrier
    uint32_t incl = s;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t v = __shfl_up(incl, off, 64);
        if (lane >= off) incl += v;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t before = 0, total = 0;
#pragma unroll
    for (int ww = 0; ww < SCAN_THREADS / 64; ww++) {
        const uint32_t v = wsum[ww];
        if (ww < wave) before += v;
        total += v;
    }
    const bool overflow = total > (uint32_t)capacity;
    uint32_t run = before + incl - s;  // exclusive prefix of this thread's chunk
    if (per <= REG) {
#pragma unroll
        for (int k = 0; k < REG; k++) {
            if (lo + k < hi) {
                // (empty tiles keep {0, 0}: the reference clears the table and writes only tiles that own instances)
                img.ranges[lo + k] = (overflow || c[k] == 0u) ? make_uint2(0u, 0u) : make_uint2(run, run + c[k]);
                run += c[k];
            }
        }
    } else {
        for (int i = lo; i < hi; i++) {
            const uint32_t cc = count_of(i);
            img.ranges[i] = (overflow || cc == 0u) ? make_uint2(0u, 0u) : make_uint2(run, run + cc);
            run += cc;
        }
    }
    if (t == 0) {
        img.status[0] = (int)total;
        img.status[1] = overflow ? 1 : 0;
        img.cursor[2] = (uint32_t)capacity;
        if (fused) {  // (otherwise scan_blocks initialised them)
            img.status[2] = (int)img.cursor[1];  // prefiltered violation
            img.status[3] = 0;                   // full variant: number of valid (pixel, Gaussian) pairs, summed by its forward blend
        }
    }
}

// One thread per Gaussian.  offset = (instances of all earlier 256-Gaussian blocks, from scan_blocks_kernel) +
// (exclusive scan inside this block).  Callback path only: the presized path counts inside preprocess_fwd.
__global__ void __launch_bounds__(256) count_rank_kernel(int P, GeometryView geom, ImageView img, BinningView bin,
                                                         int grid_x, int capacity) {
    __shared__ uint32_t wtot[4];
    __shared__ uint32_t stage[COUNT_STAGE];
    const int tid = threadIdx.x;
    const int idx = blockIdx.x * 256 + tid;
    ushort4 r = make_ushort4(0, 0, 0, 0);
    if (idx < P) r = geom.rect[idx];
    const uint32_t n = (uint32_t)(r.z - r.x) * (uint32_t)(r.w - r.y);
    uint32_t block_total;
    const uint32_t loc = block_exclusive_scan(n, wtot, tid, &block_total);
    const uint32_t block_base = geom.block_tiles[blockIdx.x];
    if (idx < P) geom.goff[idx] = block_base + loc;
    count_and_rank(r, block_base + loc, block_base, block_total, img.tile_count, bin.ranks, grid_x, capacity, stage, tid);
}

// In-place exclusive scan of the per-block instance totals (P/256 values, one 1024-thread block); the grand total
// = num_rendered goes to status[0] (the callback entry points read it before sizing the binning buffer).
__global__ void __launch_bounds__(SCAN_THREADS) scan_blocks_kernel(uint32_t* block_tiles, int nblocks, int* status) {
    __shared__ uint32_t wsum[SCAN_THREADS / 64];
    __shared__ uint32_t any_flag;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int per = (nblocks + SCAN_THREADS - 1) / SCAN_THREADS;
    const int lo = min(t * per, nblocks), hi = min(lo + per, nblocks);
    if (t == 0) any_flag = 0u;
    uint32_t s = 0, flag = 0;
    for (int i = lo; i < hi; i++) {
        const uint32_t v = block_tiles[i];  // bit 31: `prefiltered` violation seen by that block (preprocess_fwd)
        s += v & 0x7fffffffu;
        flag |= v >> 31;
    }
    uint32_t incl = s;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t v = __shfl_up(incl, off, 64);
        if (lane >= off) incl += v;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    if (flag) any_flag = 1u;
    uint32_t before = 0, total = 0;
#pragma unroll
    for (int ww = 0; ww < SCAN_THREADS / 64; ww++) {
        const uint32_t v = wsum[ww];
        if (ww < wave) before += v;
        total += v;
    }
    uint32_t run = before + incl - s;
    for (int i = lo; i < hi; i++) {
        const uint32_t c = block_tiles[i] & 0x7fffffffu;
        block_tiles[i] = run;
        run += c;
    }
    __syncthreads();
    if (t == 0) {  // the whole status word is (re)initialised here: no memset before the forward
        status[0] = (int)total;
        status[1] = 0;               // overflow: scan_tiles
        status[2] = (int)any_flag;   // prefiltered violation
        status[3] = 0;               // full variant: number of valid (pixel, Gaussian) pairs, summed by its forward blend
    }
}

// `table` (LDS count, below): slot = table[w][tile] + rank, where w is the counting workgroup of the Gaussian's
// 1024-chunk and table[w][tile] already holds range start + the instances workgroups < w counted for that tile;
// otherwise (global tile counters) slot = range start + arrival rank.
__global__ void __launch_bounds__(256) emit_instances_kernel(int P, GeometryView geom, ImageView img, BinningView bin,
                                                             int grid_x, const uint32_t* __restrict__ table, int tiles, int nwg) {
    int vb = blockIdx.x;
    if (table) {
        // Row w of the table is read by the Gaussians of the chunks w, w + nwg, ...: hand every XCD (block b runs on XCD
        // b % 8) the chunks c with c % 8 == its index -- nwg is a multiple of 8 or the whole grid is tiny -- so that a row
        // is fetched into ONE L2 (32 rows = 1 MB per XCD at 1080p) instead of into up to eight (measured: 37 -> 27 us).
        const int xcd = vb & 7, local = vb >> 3;
        vb = 4 * (xcd + 8 * (local >> 2)) + (local & 3);
    }
    const int idx = vb * 256 + threadIdx.x;
    if (idx >= P) return;
    if (img.status[1]) return;  // binning buffer too small: leave every tile list empty
    const ushort4 r = geom.rect[idx];
    if (r.z <= r.x || r.w <= r.y) return;
    const uint64_t key = ((uint64_t)__float_as_uint(geom.depths[idx]) << 32) | (uint32_t)idx;
    const uint32_t* rk = bin.ranks + geom.goff[idx];
    if (table) {
        const uint32_t* row = table + (size_t)((idx >> 10) % nwg) * tiles;  // (idx >> 10 = the Gaussian's counting chunk)
        for (int y = r.y; y < r.w; y++)
            for (int x = r.x; x < r.z; x++) bin.keys[row[y * grid_x + x] + *rk++] = key;
        return;
    }
    for (int y = r.y; y < r.w; y++)
        for (int x = r.x; x < r.z; x++) bin.keys[img.ranges[y * grid_x + x].x + *rk++] = key;
}

// ---- counting in LDS (presized path) ----------------------------------------------------------------------------
// The memory-side atomic unit retires ~26 G operations/s for the whole chip (profiles/microbench/atomics.hip): one
// returning atomic per tile instance is 55 us at config 3 and 230 us at config 4, with the CUs idle.  An LDS atomic
// costs a few cycles of ONE CU's LDS, and a whole frame's tile histogram fits one workgroup's LDS (8 160 tiles = 32 KB
// at 1080p, 32 400 = 127 KB at 3840x2160, of 160 KB).  So the instances are counted by DGR_COUNT_WGS persistent
// 1024-thread workgroups, each with a PRIVATE histogram of the whole frame in LDS:
//   count_lds  : workgroup w takes the 1024-Gaussian chunks w, w + G, ...; a Gaussian's instance run starts at
//                (instances of all earlier chunks, from preprocess_fwd's per-256-block totals) + (scan inside the chunk)
//                -- the reference's point_offsets -- and every instance takes a returning ds_add on the workgroup's
//                histogram: its rank among the instances THIS workgroup sends to that tile.  At the end the histogram
//                goes to row w of table[G][tiles] and its running sums over 64-tile segments to seg[segment][w];
//   scan_table : one workgroup per 64-tile segment: instances of all earlier segments (G values of seg), column sums = tile
//                totals -> ranges, status word; table[w][tile] <- range start + instances workgroups < w sent there;
//   emit       : slot = table[w][tile] + rank.  Any unique placement inside the tile's segment will do: sort_tiles orders it.
// No global atomics, no cleared counters, no cursor: the zero_fill launch in front of the forward is gone too.
constexpr int CL_THREADS = 1024;
constexpr int SEG_TILES = 64;

__global__ void __launch_bounds__(CL_THREADS) count_lds_kernel(int P, GeometryView geom, uint32_t* __restrict__ ranks,
                                                               uint32_t* __restrict__ table, uint32_t* __restrict__ seg,
                                                               int grid_x, int tiles, int nseg, int capacity, int prefixed) {
    // `prefixed`: geom.block_tiles holds the exclusive prefix of the 256-block totals already (callback path: scan_blocks ran)
    extern __shared__ uint32_t hist[];  // [tiles]
    __shared__ uint32_t wsum[CL_THREADS / 64], psum[CL_THREADS / 64];
    __shared__ uint32_t segsum[(DGR_COUNT_LDS_MAX_TILES + SEG_TILES - 1) / SEG_TILES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < tiles; i += CL_THREADS) hist[i] = 0u;
    const int nblocks = (P + 255) / 256, nchunks = (P + CL_THREADS - 1) / CL_THREADS;
    uint32_t run = 0;    // instances of the 256-blocks [0, next_block)
    int next_block = 0;
    __syncthreads();
    for (int chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        // instances of the chunks between the previous one of this workgroup and this one
        const int first = 4 * chunk;
        uint32_t part = 0;
        if (!prefixed)
            for (int b = next_block + tid; b < first; b += CL_THREADS) part += geom.block_tiles[b] & 0x7fffffffu;
        const int idx = chunk * CL_THREADS + tid;
        ushort4 r = make_ushort4(0, 0, 0, 0);
        if (idx < P) r = geom.rect[idx];
        const uint32_t w = (uint32_t)(r.z - r.x), h = (uint32_t)(r.w - r.y);
        const uint32_t n = w * h;
        uint32_t incl = n;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t v = __shfl_up(incl, off, 64);
            if (lane >= off) incl += v;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
        if (lane == 63) wsum[wave] = incl;
        if (lane == 0) psum[wave] = part;
        __syncthreads();
        uint32_t before = 0, chunk_total = 0, skipped = 0;
#pragma unroll
        for (int ww = 0; ww < CL_THREADS / 64; ww++) {
            const uint32_t v = wsum[ww];
            if (ww < wave) before += v;
            chunk_total += v;
            skipped += psum[ww];
        }
        __syncthreads();  // (wsum / psum are rewritten by the next chunk)
        if (prefixed) run = geom.block_tiles[first];
        const uint32_t off0 = run + skipped + before + incl - n;
        run += skipped + chunk_total;
        next_block = min(first + 4, nblocks);
        if (idx < P) geom.goff[idx] = off0;
        // past the capacity an instance is still counted, so that scan_table sees the true total and flags the overflow
        const bool store = (unsigned long long)off0 + n <= (unsigned long long)(uint32_t)capacity;
        uint32_t k = off0;
        for (uint32_t y = r.y; y < r.w; y++) {
            uint32_t* hrow = hist + y * (uint32_t)grid_x;
            for (uint32_t x = r.x; x < r.z; x++, k++) {
                const uint32_t rank = atomicAdd(hrow + x, 1u);
                if (store) ranks[k] = rank;
            }
        }
    }
    __syncthreads();
    uint32_t* row = table + (size_t)blockIdx.x * tiles;
    for (int i = tid; i < tiles; i += CL_THREADS) row[i] = hist[i];
    // seg[s][w] = instances this workgroup sent to the tiles of segments 0..s (inclusive scan over the segments)
    for (int sgm = wave; sgm < nseg; sgm += CL_THREADS / 64) {
        const int t = sgm * SEG_TILES + lane;
        uint32_t v = (t < tiles) ? hist[t] : 0u;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if (lane == 0) segsum[sgm] = v;
    }
    __syncthreads();
    if (wave == 0) {
        uint32_t carry = 0;
        for (int b0 = 0; b0 < nseg; b0 += 64) {
            uint32_t incl = (b0 + lane < nseg) ? segsum[b0 + lane] : 0u;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t v = __shfl_up(incl, off, 64);
                if (lane >= off) incl += v;
            }
            if (b0 + lane < nseg) seg[(size_t)(b0 + lane) * gridDim.x + blockIdx.x] = carry + incl;
            carry += __shfl(incl, 63, 64);
        }
    }
}

// One workgroup per 64-tile segment; wave g handles the table rows [g * rows_per, (g + 1) * rows_per) of the segment's
// 64 columns (lane = tile: 256-byte row pieces).  nwg <= 16 * ROWS_MAX.
constexpr int ST_ROWS_MAX = 16;
__global__ void __launch_bounds__(CL_THREADS) scan_table_kernel(ImageView img, uint32_t* __restrict__ table,
                                                                const uint32_t* __restrict__ seg,
                                                                const uint32_t* __restrict__ block_tiles, int nblocks,
                                                                int tiles, int nwg, int capacity, int prefixed) {
    __shared__ uint32_t grp[CL_THREADS / 64][SEG_TILES];
    __shared__ uint32_t red[3][CL_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sgm = blockIdx.x;
    const int tile = sgm * SEG_TILES + lane;
    const bool live = tile < tiles;
    const int rows_per = (nwg + CL_THREADS / 64 - 1) / (CL_THREADS / 64);
    const int r0 = wave * rows_per;
    // Order of the workgroups' runs inside a tile's segment: XCD-major.  Counting workgroup w ran on XCD w % 8 and emit
    // hands its Gaussians to the same XCD, so with the runs of one XCD adjacent every cache line of the key array is
    // written by ONE XCD's L2 and leaves it complete -- with the runs in plain w order each line collected 8-byte pieces
    // in eight L2s and went to memory eight times (emit: 57 MB written for 13 MB of keys).  Position p of the order is
    // workgroup (p % (nwg / 8)) * 8 + p / (nwg / 8); any order is a valid placement.
    const int per_xcd = nwg >> 3;
    const bool xcd_major = (nwg & 7) == 0;
    int rows[ST_ROWS_MAX];  // (one division per wave, then incremental: the positions of a wave are consecutive)
    {
        int q = xcd_major ? r0 % per_xcd : 0, x = xcd_major ? r0 / per_xcd : 0;
#pragma unroll
        for (int i = 0; i < ST_ROWS_MAX; i++) {
            rows[i] = xcd_major ? q * 8 + x : r0 + i;
            if (++q == per_xcd) { q = 0; x++; }
        }
    }
    uint32_t c[ST_ROWS_MAX];
    uint32_t mine = 0;
#pragma unroll
    for (int i = 0; i < ST_ROWS_MAX; i++) {
        c[i] = (live && i < rows_per && r0 + i < nwg) ? table[(size_t)rows[i] * tiles + tile] : 0u;
        mine += c[i];
    }
    grp[wave][lane] = mine;
    // instances of all earlier segments and the grand total, from the workgroups' inclusive segment sums; block 0 also
    // collects the `prefiltered` flag from preprocess_fwd's block totals
    uint32_t before_seg = 0, total = 0, flag = 0;
    const int nseg = gridDim.x;
    for (int i = tid; i < nwg; i += CL_THREADS) {
        if (sgm > 0) before_seg += seg[(size_t)(sgm - 1) * nwg + i];
        total += seg[(size_t)(nseg - 1) * nwg + i];
    }
    if (sgm == 0 && !prefixed)  // (callback path: scan_blocks has moved the flag into status[2] already)
        for (int i = tid; i < nblocks; i += CL_THREADS) flag |= block_tiles[i] >> 31;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        before_seg += __shfl_xor(before_seg, off, 64);
        total += __shfl_xor(total, off, 64);
        flag |= __shfl_xor(flag, off, 64);
    }
    if (lane == 0) { red[0][wave] = before_seg; red[1][wave] = total; red[2][wave] = flag; }
    __syncthreads();
    before_seg = 0; total = 0; flag = 0;
#pragma unroll
    for (int ww = 0; ww < CL_THREADS / 64; ww++) { before_seg += red[0][ww]; total += red[1][ww]; flag |= red[2][ww]; }
    uint32_t above = 0, count = 0;  // instances the row groups before this wave's sent to the tile; the tile's total
#pragma unroll
    for (int ww = 0; ww < CL_THREADS / 64; ww++) {
        const uint32_t v = grp[ww][lane];
        if (ww < wave) above += v;
        count += v;
    }
    uint32_t incl = count;  // exclusive scan over the segment's 64 tiles (every wave computes the same)
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t v = __shfl_up(incl, off, 64);
        if (lane >= off) incl += v;
    }
    const bool overflow = total > (uint32_t)capacity;
    const uint32_t start = before_seg + incl - count;
    uint32_t base = start + above;
#pragma unroll
    for (int i = 0; i < ST_ROWS_MAX; i++) {
        if (live && i < rows_per && r0 + i < nwg) table[(size_t)rows[i] * tiles + tile] = base;
        base += c[i];
    }
    // (empty tiles keep {0, 0}: the reference clears the table and writes only tiles that own instances)
    if (wave == 0 && live) img.ranges[tile] = (overflow || count == 0u) ? make_uint2(0u, 0u) : make_uint2(start, start + count);
    if (sgm == 0 && tid == 0) {
        img.status[0] = (int)total;
        img.status[1] = overflow ? 1 : 0;
        if (!prefixed) {
            img.status[2] = (int)flag;  // prefiltered violation
            img.status[3] = 0;          // full variant: number of valid (pixel, Gaussian) pairs, summed by its forward blend
        }
        img.cursor[2] = (uint32_t)capacity;
    }
}

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

__global__ void __launch_bounds__(SORT_THREADS) sort_tiles_kernel(ImageView img, BinningView bin) {
    __shared__ uint64_t sk[SORT_LDS_MAX];
    const int tile = blockIdx.x;
    const uint2 rg = img.ranges[tile];
    const int n = (int)(rg.y - rg.x);
    if (n <= 0) return;
    uint64_t* gk = bin.keys + rg.x;
    uint32_t* pl = bin.point_list + rg.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (n > SORT_LDS_MAX) {  // oversize tile: same network, in place in global memory, one barrier per step
        int np2 = 1;
        while (np2 < n) np2 <<= 1;
        const int half = np2 >> 1;
        for (int size = 2; size <= np2; size <<= 1) {
            const int hs = size >> 1;
            for (int t = tid; t < half; t += SORT_THREADS) {
                const int blk = t / hs, off = t - blk * hs;
                const int i = blk * size + off, j = blk * size + (size - 1 - off);
                if (j < n) { const uint64_t x = gk[i], y = gk[j]; if (x > y) { gk[i] = y; gk[j] = x; } }
            }
            __syncthreads();
            for (int d = size >> 2; d > 0; d >>= 1) {
                for (int t = tid; t < half; t += SORT_THREADS) {
                    const int blk = t / d, off = t - blk * d;
                    const int i = blk * 2 * d + off, j = i + d;
                    if (j < n) { const uint64_t x = gk[i], y = gk[j]; if (x > y) { gk[i] = y; gk[j] = x; } }
                }
                __syncthreads();
            }
        }
        for (int i = tid; i < n; i += SORT_THREADS) pl[i] = (uint32_t)gk[i];
        return;
    }
    int np2 = 64;
    while (np2 < n) np2 <<= 1;
    const int chunks = np2 >> 6;
    // load + phase A: every 64-chunk sorted in registers
    for (int c = wave; c < chunks; c += 4) {
        const int e = c * 64 + lane;
        uint64_t v = (e < n) ? gk[e] : KEY_INF;
        sk[e] = chunk_sort64(v, lane);
    }
    __syncthreads();
    const int half = np2 >> 1;
    for (int size = 128; size <= np2; size <<= 1) {
        {  // flip across the `size` block (distance >= 64 for every pair once the chunks are sorted ... not always:
           // pairs i <-> blk*size + size-1-off span all distances, so this step runs on the LDS array)
            const int hs = size >> 1;
            for (int t = tid; t < half; t += SORT_THREADS) {
                const int blk = t / hs, off = t - blk * hs;
                const int i = blk * size + off, j = blk * size + (size - 1 - off);
                const uint64_t x = sk[i], y = sk[j];
                if (x > y) { sk[i] = y; sk[j] = x; }
            }
            __syncthreads();
        }
        for (int d = size >> 2; d >= 64; d >>= 1) {
            for (int t = tid; t < half; t += SORT_THREADS) {
                const int blk = t / d, off = t - blk * d;
                const int i = blk * 2 * d + off, j = i + d;
                const uint64_t x = sk[i], y = sk[j];
                if (x > y) { sk[i] = y; sk[j] = x; }
            }
            __syncthreads();
        }
        for (int c = wave; c < chunks; c += 4) {
            const int e = c * 64 + lane;
            sk[e] = chunk_tail64(sk[e], lane);
        }
        __syncthreads();
    }
    for (int i = tid; i < n; i += SORT_THREADS) pl[i] = (uint32_t)sk[i];
}

}  // namespace

hipError_t launch_scan_tiles(ImageView img, int tiles, int grid_x, int capacity, bool fused, hipStream_t stream) {
    launch(scan_tiles_kernel, dim3(1), dim3(SCAN_THREADS), stream, img, tiles, grid_x, capacity, fused ? 1 : 0);
    return hipGetLastError();
}
hipError_t launch_count_rank(int P, GeometryView geom, ImageView img, BinningView bin, int grid_x, int capacity,
                             hipStream_t stream) {
    if (P <= 0) return hipSuccess;
    launch(count_rank_kernel, dim3((P + 255) / 256), dim3(256), stream, P, geom, img, bin, grid_x, capacity);
    return hipGetLastError();
}
hipError_t launch_scan_blocks(int P, GeometryView geom, ImageView img, hipStream_t stream) {
    launch(scan_blocks_kernel, dim3(1), dim3(SCAN_THREADS), stream, geom.block_tiles, (P + 255) / 256, img.status);
    return hipGetLastError();
}
hipError_t launch_emit_instances(int P, GeometryView geom, ImageView img, BinningView bin, int grid_x, hipStream_t stream,
                                 const uint32_t* table, int tiles, int nwg) {
    if (P <= 0) return hipSuccess;
    // with a table: 8 XCDs x (chunks per XCD) x 4 blocks per 1024-chunk (blocks past P return at once)
    const int nblocks = table ? 8 * 4 * (((P + 1023) / 1024 + 7) / 8) : (P + 255) / 256;
    launch(emit_instances_kernel, dim3(nblocks), dim3(256), stream, P, geom, img, bin, grid_x, table, tiles, nwg);
    return hipGetLastError();
}
int count_lds_workgroups(int P) { return max(1, min(DGR_COUNT_WGS, (P + CL_THREADS - 1) / CL_THREADS)); }
bool count_lds_fits(int tiles) { return tiles > 0 && tiles <= DGR_COUNT_LDS_MAX_TILES; }
hipError_t launch_count_lds(int P, GeometryView geom, BinningView bin, CountTable ct, int grid_x, int tiles, int capacity,
                            bool prefixed, hipStream_t stream) {
    static std::once_flag once;
    static hipError_t attr_rc = hipSuccess;
    std::call_once(once, [] {
        attr_rc = hipFuncSetAttribute(reinterpret_cast<const void*>(count_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      DGR_COUNT_LDS_MAX_TILES * 4);
    });
    if (attr_rc != hipSuccess) return attr_rc;
    const int nwg = count_lds_workgroups(P), nseg = (tiles + SEG_TILES - 1) / SEG_TILES;
    launch_shmem(count_lds_kernel, dim3(nwg), dim3(CL_THREADS), (size_t)tiles * 4, stream, P, geom, bin.ranks, ct.table, ct.seg,
                 grid_x, tiles, nseg, capacity, prefixed ? 1 : 0);
    return hipGetLastError();
}
hipError_t launch_scan_table(int P, GeometryView geom, ImageView img, CountTable ct, int tiles, int capacity, bool prefixed,
                             hipStream_t stream) {
    const int nwg = count_lds_workgroups(P), nseg = (tiles + SEG_TILES - 1) / SEG_TILES;
    launch(scan_table_kernel, dim3(nseg), dim3(CL_THREADS), stream, img, ct.table, ct.seg, geom.block_tiles, (P + 255) / 256,
           tiles, nwg, capacity, prefixed ? 1 : 0);
    return hipGetLastError();
}
hipError_t launch_sort_tiles(ImageView img, BinningView bin, int tiles, hipStream_t stream) {
    launch(sort_tiles_kernel, dim3(tiles), dim3(SORT_THREADS), stream, img, bin);
    return hipGetLastError();
}

}  // namespace dgr
