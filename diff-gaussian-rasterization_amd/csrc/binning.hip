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

__global__ void __launch_bounds__(SCAN_THREADS) scan_tiles_kernel(ImageView img, int tiles, int grid_x, int capacity, int fused,
                                                                  int sched_on, StatusReport rep) {
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
        // bit 0: the blend kernels walk tile_sched; bit 1: this frame overflowed; bit 2: quadrant lists in the light blend kernels
        // (segment_binning.hip decides that per frame; this path has no run statistics and takes only the forced setting)
        img.cursor[3] = (uint32_t)(sched_on & 1) | (overflow ? 2u : 0u) | ((sched_on & BLEND_LISTS_QUADRANT) ? 4u : 0u);
        if (fused) {  // (otherwise scan_blocks initialised them)
            img.status[2] = (int)img.cursor[1];  // prefiltered violation
            img.status[3] = 0;                   // full variant: number of valid (pixel, Gaussian) pairs, summed by its forward blend
        }
        // (this path does not track the longest list: "unknown" keeps the tile schedule on)
        if (rep.ws) __hip_atomic_store(rep.ws, 0x7fffffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
// heaviest-first takes 124 / 220 (profiles/r6/tile_order_clustered.txt).  One workgroup writes, per workgroup slot b of the
// blend kernels, {tile, list start, list end} (they need no second lookup):
//   * the tiles go by CLASS of list length, longest first -- two classes per octave (lengths within ~40 % of each other share
//     one), which is all the balance needs: what matters is that a 1000-entry list does not start behind 200-entry ones;
//   * inside a class the tiles keep their order in the image, and the class is dealt to the
//     XCDs in eight CONTIGUOUS parts -- slot b runs on XCD b mod 8 -- so that tiles running together on an XCD are neighbours
//     and share the Gaussians' records in its L2.  (A fine sort by length scatters neighbours over the XCDs: same kernel
//     times, but 350 / 359 MB of HBM traffic for the forward / backward blend instead of 188 / 226, profiles/r6_pmc.txt.)
// On the uniform benchmark scene nine tiles in ten share one class: the schedule is round 5's XCD bands with the few long lists
// in front.
constexpr int TS_THREADS = 1024;
// A class owns the M slots from B on; XCD x = slot mod 8 takes the x-th contiguous part of the class, as many tiles as the class
// has slots on that XCD.  part_table fills, for one (class, XCD): the class position its part starts at and its first slot.
__device__ __forceinline__ void part_table(uint32_t B, uint32_t M, uint32_t x, uint32_t& n_x, uint32_t& first) {
    first = B + ((x + 8u - (B & 7u)) & 7u);  // first slot >= B on XCD x
    n_x = first < B + M ? (B + M - 1u - first) / 8u + 1u : 0u;
}
// Wave w takes the w-th contiguous sixteenth of the tiles, 512 at a time, EIGHT CONSECUTIVE TILES PER LANE: neighbours mostly
// share a class, so a lane hands in whole runs -- one LDS atomic per run instead of one per tile (same-address LDS atomics retire
// about one lane per two cycles for the whole CU: 2 x 8160 of them on the uniform scene's one dominant class were 13 of a first
// version's 16 us).  First every wave counts its tiles per class in its own row of counters; a prefix over the waves turns the
// rows into each wave's first position inside every class; then every wave places its tiles, drawing positions from its own
// row: no barrier between the waves, a class's tiles in image order up to the order of the lanes inside one 512-tile block.
__global__ void __launch_bounds__(TS_THREADS) tile_schedule_kernel(const uint2* __restrict__ ranges, int tiles,
                                                                    uint4* __restrict__ sched) {
    constexpr int NW = TS_THREADS / 64, NC = DGR_SCHED_CLASSES, PL = 8;
    __shared__ uint32_t cntw[NW][NC];  // tiles of wave w in class c; then: position of the wave's next tile inside class c
    __shared__ uint32_t cnt[NC], base[NC];
    __shared__ uint32_t part_off[NC][8], part_first[NC][8];  // per (class, XCD): first class position / first slot of the part
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < NW * NC; i += TS_THREADS) (&cntw[0][0])[i] = 0u;
    __syncthreads();
    const int per_wave = ((tiles + NW - 1) / NW + 64 * PL - 1) / (64 * PL) * (64 * PL);  // a multiple of 512
    const int w0 = wave * per_wave, w1 = min(tiles, w0 + per_wave);
    // the lane's eight tiles of the block starting at b0: ranges, classes, and the runs of equal class among them
    struct Block { uint2 r[PL]; uint32_t c[PL]; uint32_t len[PL]; bool valid[PL], start[PL]; };
    auto load_block = [&](int b0, Block& B) {
        const int t0 = b0 + lane * PL;
        const uint4* src = reinterpret_cast<const uint4*>(ranges + t0);  // (16-byte aligned: t0 is a multiple of 8; a block's tail may lie
#pragma unroll                                                          //  behind `tiles` but inside the image buffer: ignored)
        for (int k = 0; k < PL; k += 2) {
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (t0 + k < w1) v = src[k / 2];
            B.r[k] = make_uint2(v.x, v.y);
            B.r[k + 1] = make_uint2(v.z, v.w);
        }
#pragma unroll
        for (int k = 0; k < PL; k++) {
            B.valid[k] = t0 + k < w1;
            B.c[k] = sched_class(B.r[k].y - B.r[k].x);
            B.start[k] = B.valid[k] && (k == 0 || B.c[k] != B.c[k - 1]);
        }
        B.len[PL - 1] = 1u;
#pragma unroll
        for (int k = PL - 2; k >= 0; k--) B.len[k] = (B.valid[k + 1] && B.c[k + 1] == B.c[k]) ? B.len[k + 1] + 1u : 1u;  // run length from k on
    };
    for (int b0 = w0; b0 < w1; b0 += 64 * PL) {
        Block B;
        load_block(b0, B);
#pragma unroll
        for (int k = 0; k < PL; k++)
            if (B.start[k]) atomicAdd(&cntw[wave][B.c[k]], B.len[k]);
    }
    __syncthreads();
    {   // every wave's first position inside each class (thread (w, c) sums the rows above its own), the class totals
        const int w = tid >> 5, c = tid & (NC - 1);
        uint32_t before = 0u;
        if (tid < NW * NC)
            for (int ww = 0; ww < w; ww++) before += cntw[ww][c];
        if (tid >= (NW - 1) * NC && tid < NW * NC) cnt[c] = before + cntw[NW - 1][c];
        __syncthreads();
        if (tid < NW * NC) cntw[w][c] = before;
    }
    __syncthreads();
    if (tid < 64) {  // first slot of every class, the class of the longest lists first (lane l holds class NC - 1 - l)
        const uint32_t v = lane < NC ? cnt[NC - 1 - lane] : 0u;
        uint32_t incl = v;
#pragma unroll
        for (int off = 1; off < NC; off <<= 1) {
            const uint32_t t = __shfl_up(incl, off, 64);
            if (lane >= off) incl += t;
        }
        if (lane < NC) base[NC - 1 - lane] = incl - v;
    }
    __syncthreads();
    if (tid < NC * 8) {  // the XCDs' parts of every class
        const uint32_t c = (uint32_t)tid >> 3, x = (uint32_t)tid & 7u;
        uint32_t off = 0u, n_x, first;
        for (uint32_t xx = 0; xx < x; xx++) { part_table(base[c], cnt[c], xx, n_x, first); off += n_x; }
        part_table(base[c], cnt[c], x, n_x, first);
        part_off[c][x] = off;
        part_first[c][x] = first;
    }
    __syncthreads();
    for (int b0 = w0; b0 < w1; b0 += 64 * PL) {
        Block B;
        load_block(b0, B);
        uint32_t run_base = 0u;
#pragma unroll
        for (int k = 0; k < PL; k++) {
            if (B.start[k]) run_base = atomicAdd(&cntw[wave][B.c[k]], B.len[k]);  // (the wave's own row)
            else run_base += 1u;                                                     // (next tile of the same run)
            if (B.valid[k]) {
                const uint32_t c = B.c[k], p = run_base;  // position inside the class
                uint32_t x = 0u;  // the part holding it: the last one that starts at or before p (an empty part starts where the next does)
#pragma unroll
                for (int j = 1; j < 8; j++) x += p >= part_off[c][j] ? 1u : 0u;
                sched[part_first[c][x] + 8u * (p - part_off[c][x])] = make_uint4((uint32_t)(b0 + lane * PL + k), B.r[k].x, B.r[k].y, 0u);
            }
        }
    }
}

}  // namespace

hipError_t launch_tile_schedule(ImageView img, int tiles, hipStream_t stream) {
    if (tiles <= 0) return hipSuccess;
    launch(tile_schedule_kernel, dim3(1), dim3(TS_THREADS), stream, (const uint2*)img.ranges, tiles, img.tile_sched);
    return hipGetLastError();
}

hipError_t launch_scan_tiles(ImageView img, int tiles, int grid_x, int capacity, bool fused, int blend_flags, StatusReport rep,
                             hipStream_t stream) {
    launch(scan_tiles_kernel, dim3(1), dim3(SCAN_THREADS), stream, img, tiles, grid_x, capacity, fused ? 1 : 0, blend_flags, rep);
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
