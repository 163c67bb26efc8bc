// binning.hip -- instance binning for gfx950: tile histogram -> ranges, per-tile key emission,
// per-tile sort.
//
// Replaces cub::DeviceScan::InclusiveSum over P, duplicateWithKeys, cub::DeviceRadixSort::SortPairs
// on 64-bit (tile | depth) keys and identifyTileRanges (L/cuda_rasterizer/rasterizer_impl.cu:70-138,
// 283-323).  The reference sorts all R instances globally on (tile id, depth bits) with a STABLE
// radix sort whose input is in ascending Gaussian order, so its result is the unique ascending
// order on the triple (tile id, depth bits, gaussian id).  Here the tile id never enters a key:
//   1. preprocess already histogrammed instances per tile (tile_count);
//   2. scan_tiles turns the histogram into the range table (this IS identifyTileRanges' output);
//   3. emit_instances drops (depth bits << 32 | gaussian id) keys into their tile's segment in
//      arbitrary order (one returning atomic per instance);
//   4. sort_tiles sorts each segment in LDS -- a total order on unique keys, hence the same
//      point_list as the reference bit for bit.
// HBM traffic: 8 B written + 8 B read + 12 B written per instance, against ~6 radix passes x 24 B
// in the reference.
#include "dgr_common.h"
#include "kernels.h"

namespace dgr {
namespace {

constexpr int SCAN_THREADS = 1024;
constexpr int SORT_THREADS = 256;
constexpr int SORT_LDS_MAX = 4096;  // keys per tile sorted in LDS (32 KB); larger tiles sort in global memory

__global__ void __launch_bounds__(SCAN_THREADS) scan_tiles_kernel(ImageView img, int tiles, int capacity) {
    __shared__ uint32_t part[SCAN_THREADS];
    const int t = threadIdx.x;
    const int per = (tiles + SCAN_THREADS - 1) / SCAN_THREADS;
    const int lo = min(t * per, tiles), hi = min(lo + per, tiles);
    uint32_t s = 0;
    for (int i = lo; i < hi; i++) s += img.tile_count[i];
    part[t] = s;
    __syncthreads();
    // Hillis-Steele inclusive scan over the 1024 partials
    for (int off = 1; off < SCAN_THREADS; off <<= 1) {
        uint32_t v = (t >= off) ? part[t - off] : 0u;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    const uint32_t total = part[SCAN_THREADS - 1];
    const bool overflow = total > (uint32_t)capacity;
    uint32_t run = part[t] - s;  // exclusive prefix of this thread's chunk
    for (int i = lo; i < hi; i++) {
        const uint32_t c = img.tile_count[i];
        img.ranges[i] = overflow ? make_uint2(0u, 0u) : make_uint2(run, run + c);
        img.tile_fill[i] = 0u;
        run += c;
    }
    if (t == 0) {
        img.status[0] = (int)total;
        img.status[1] = overflow ? 1 : 0;
    }
}

__global__ void __launch_bounds__(256) emit_instances_kernel(int P, GeometryView geom, ImageView img, BinningView bin,
                                                             int grid_x) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= P) return;
    if (img.status[1]) return;  // binning buffer too small: leave every tile list empty
    const ushort4 r = geom.rect[idx];
    if (r.z <= r.x || r.w <= r.y) return;
    const uint64_t key = ((uint64_t)__float_as_uint(geom.depths[idx]) << 32) | (uint32_t)idx;
    for (int y = r.y; y < r.w; y++)
        for (int x = r.x; x < r.z; x++) {
            const int tile = y * grid_x + x;
            const uint32_t slot = img.ranges[tile].x + atomicAdd(&img.tile_fill[tile], 1u);
            bin.keys[slot] = key;
        }
}

// All-ascending bitonic network ("flip" then "disperse" steps).  Every comparator moves the smaller
// key to the lower index, so indices >= n behave as +inf padding and comparators that touch them are
// skipped: no power-of-two padding is materialised.
template <typename KeyPtr>
__device__ __forceinline__ void bitonic_sort(KeyPtr k, int n, int tid) {
    int np2 = 1;
    while (np2 < n) np2 <<= 1;
    const int half = np2 >> 1;
    for (int size = 2; size <= np2; size <<= 1) {
        // flip: i in the lower half of its `size` block pairs with the mirrored index
        {
            const int hs = size >> 1;
            for (int t = tid; t < half; t += SORT_THREADS) {
                const int blk = t / hs, off = t - blk * hs;
                const int i = blk * size + off, j = blk * size + (size - 1 - off);
                if (j < n) {
                    const uint64_t a = k[i], b = k[j];
                    if (a > b) { k[i] = b; k[j] = a; }
                }
            }
            __syncthreads();
        }
        for (int d = size >> 2; d > 0; d >>= 1) {
            for (int t = tid; t < half; t += SORT_THREADS) {
                const int blk = t / d, off = t - blk * d;
                const int i = blk * 2 * d + off, j = i + d;
                if (j < n) {
                    const uint64_t a = k[i], b = k[j];
                    if (a > b) { k[i] = b; k[j] = a; }
                }
            }
            __syncthreads();
        }
    }
}

__global__ void __launch_bounds__(SORT_THREADS) sort_tiles_kernel(ImageView img, BinningView bin) {
    __shared__ uint64_t sk[SORT_LDS_MAX];
    const int tile = blockIdx.x;
    const uint2 rg = img.ranges[tile];
    const int n = (int)(rg.y - rg.x);
    if (n <= 0) return;
    uint64_t* gk = bin.keys + rg.x;
    uint32_t* pl = bin.point_list + rg.x;
    const int tid = threadIdx.x;
    if (n <= SORT_LDS_MAX) {
        for (int i = tid; i < n; i += SORT_THREADS) sk[i] = gk[i];
        __syncthreads();
        bitonic_sort(sk, n, tid);
        for (int i = tid; i < n; i += SORT_THREADS) {
            const uint64_t v = sk[i];
            gk[i] = v;
            pl[i] = (uint32_t)v;
        }
    } else {
        __syncthreads();
        bitonic_sort(gk, n, tid);
        for (int i = tid; i < n; i += SORT_THREADS) pl[i] = (uint32_t)gk[i];
    }
}

}  // namespace

hipError_t launch_scan_tiles(ImageView img, int tiles, int capacity, hipStream_t stream) {
    hipLaunchKernelGGL(scan_tiles_kernel, dim3(1), dim3(SCAN_THREADS), 0, stream, img, tiles, capacity);
    return hipGetLastError();
}
hipError_t launch_emit_instances(int P, GeometryView geom, ImageView img, BinningView bin, int grid_x, hipStream_t stream) {
    if (P <= 0) return hipSuccess;
    hipLaunchKernelGGL(emit_instances_kernel, dim3((P + 255) / 256), dim3(256), 0, stream, P, geom, img, bin, grid_x);
    return hipGetLastError();
}
hipError_t launch_sort_tiles(ImageView img, BinningView bin, int tiles, hipStream_t stream) {
    hipLaunchKernelGGL(sort_tiles_kernel, dim3(tiles), dim3(SORT_THREADS), 0, stream, img, bin);
    return hipGetLastError();
}

}  // namespace dgr
