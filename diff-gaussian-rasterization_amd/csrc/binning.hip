// binning.hip -- instance binning on global tile counters (gfx950): the path of frames whose segment tables do not fit
// LDS and of dgr_set_option("lds_count", 0); everything else goes through segment_binning.hip.
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
// address spread or whether a value is returned (profiles/microbench/atomics.hip), i.e. ~63 us per 1.65 M.
#include "dgr_common.h"
#include "kernels.h"
#include "count_rank.h"
#include "tile_sort.h"
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

// slot = range start + arrival rank (count_rank's returning atomics on the global tile counters)
__global__ void __launch_bounds__(256) emit_instances_kernel(int P, GeometryView geom, ImageView img, BinningView bin, int grid_x) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= P) return;
    if (img.status[1]) return;  // binning buffer too small: leave every tile list empty
    const ushort4 r = geom.rect[idx];
    if (r.z <= r.x || r.w <= r.y) return;
    const uint64_t key = ((uint64_t)__float_as_uint(geom.depths[idx]) << 32) | (uint32_t)idx;
    const uint32_t* rk = bin.ranks + geom.goff[idx];
    for (int y = r.y; y < r.w; y++)
        for (int x = r.x; x < r.z; x++) bin.keys[img.ranges[y * grid_x + x].x + *rk++] = key;
}

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

// ---- tile schedule ------------------------------------------------------------------------------------------------
// The blend kernels run one workgroup per tile and a workgroup's time grows with its list.  On a scene whose Gaussians
// cluster, a static block -> tile map hands whole clusters to a few XCDs and starts the longest lists last: on a frame with
// lists of 51 .. 1135 entries (mean 222) the XCD-band map of rounds 1-5 took 236 / 445 us (forward / backward blend) where
// heaviest-first takes 124 / 220 (profiles/r6/tile_order_clustered.txt).  One workgroup orders the tiles by descending list
// length with a counting sort on 256 buckets (bucket = length >> shift, shift from the longest list) and writes, per
// workgroup slot b of the blend kernels, {tile, list start, list end}: the blend kernels then need no second lookup.
// Inside a bucket the order is the arrival order of LDS atomics (only scheduling depends on it).
constexpr int TS_THREADS = 1024;
__global__ void __launch_bounds__(TS_THREADS) tile_schedule_kernel(const uint2* __restrict__ ranges, int tiles,
                                                                    uint4* __restrict__ sched) {
    __shared__ uint32_t hist[256];
    __shared__ uint32_t wtot[4];
    __shared__ uint32_t s_max;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < 256) hist[tid] = 0u;
    if (tid == 0) s_max = 0u;
    __syncthreads();
    constexpr int KEEP = 8;  // ranges kept in registers between the passes (frames up to 8192 tiles need no second read)
    uint2 rg[KEEP];
    uint32_t m = 0u;
#pragma unroll
    for (int k = 0; k < KEEP; k++) {
        const int t = tid + k * TS_THREADS;
        rg[k] = t < tiles ? ranges[t] : make_uint2(0u, 0u);
        m = max(m, rg[k].y - rg[k].x);
    }
    for (int t = tid + KEEP * TS_THREADS; t < tiles; t += TS_THREADS) { const uint2 r = ranges[t]; m = max(m, r.y - r.x); }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, off, 64));
    if (lane == 0 && m) atomicMax(&s_max, m);
    __syncthreads();
    const int shift = max(0, 24 - (int)__clz(s_max | 1u));  // (longest list) >> shift < 256
#pragma unroll
    for (int k = 0; k < KEEP; k++)
        if (tid + k * TS_THREADS < tiles) atomicAdd(&hist[255u - ((rg[k].y - rg[k].x) >> shift)], 1u);
    for (int t = tid + KEEP * TS_THREADS; t < tiles; t += TS_THREADS) { const uint2 r = ranges[t]; atomicAdd(&hist[255u - ((r.y - r.x) >> shift)], 1u); }
    __syncthreads();
    // exclusive scan of the 256 bucket counts (bucket 0 = the longest lists)
    uint32_t n = 0u, incl = 0u;
    if (tid < 256) {
        n = hist[tid];
        incl = n;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t v = __shfl_up(incl, off, 64);
            if (lane >= off) incl += v;
        }
        if (lane == 63) wtot[wave] = incl;
    }
    __syncthreads();
    if (tid < 256) {
        uint32_t before = 0u;
        for (int w = 0; w < wave; w++) before += wtot[w];
        hist[tid] = before + incl - n;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < KEEP; k++) {
        const int t = tid + k * TS_THREADS;
        if (t < tiles) {
            const uint32_t pos = atomicAdd(&hist[255u - ((rg[k].y - rg[k].x) >> shift)], 1u);
            sched[pos] = make_uint4((uint32_t)t, rg[k].x, rg[k].y, 0u);
        }
    }
    for (int t = tid + KEEP * TS_THREADS; t < tiles; t += TS_THREADS) {
        const uint2 r = ranges[t];
        const uint32_t pos = atomicAdd(&hist[255u - ((r.y - r.x) >> shift)], 1u);
        sched[pos] = make_uint4((uint32_t)t, r.x, r.y, 0u);
    }
}

}  // namespace

hipError_t launch_tile_schedule(ImageView img, int tiles, hipStream_t stream) {
    if (tiles <= 0) return hipSuccess;
    launch(tile_schedule_kernel, dim3(1), dim3(TS_THREADS), stream, (const uint2*)img.ranges, tiles, img.tile_sched);
    return hipGetLastError();
}

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
hipError_t launch_emit_instances(int P, GeometryView geom, ImageView img, BinningView bin, int grid_x, hipStream_t stream) {
    if (P <= 0) return hipSuccess;
    launch(emit_instances_kernel, dim3((P + 255) / 256), dim3(256), stream, P, geom, img, bin, grid_x);
    return hipGetLastError();
}
hipError_t launch_sort_tiles(ImageView img, BinningView bin, int tiles, hipStream_t stream) {
    launch(sort_tiles_kernel, dim3(tiles), dim3(SORT_THREADS), stream, img, bin);
    return hipGetLastError();
}

}  // namespace dgr
