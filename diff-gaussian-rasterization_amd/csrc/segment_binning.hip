// segment_binning.hip -- two-level instance binning for gfx950 (presized path and callback path; frames whose segment
// tables fit LDS).
//
// Replaces cub::DeviceScan::InclusiveSum over P, duplicateWithKeys, cub::DeviceRadixSort::SortPairs on 64-bit
// (tile | depth) keys and identifyTileRanges (L/cuda_rasterizer/rasterizer_impl.cu:70-138, 283-323) with TWO kernels and
// no global atomics.  The reference's result -- point_list ascending on (tile id, depth bits, gaussian id), ranges per tile --
// is reproduced bit for bit: keys are unique, so any placement followed by a sort of each tile's list yields it.
//
// Round 2-4 placed every tile INSTANCE individually: a table of per-(counting workgroup, tile) slot bases (8 MB at 1080p),
// one 8-byte key scattered per instance (each reaching memory as its own 32-byte write: 4.6x write amplification), a
// per-tile sort reading the keys back -- count_lds + scan_table + emit + sort_tiles = 66 us at config 3.  Here a
// Gaussian is first binned by ROW SEGMENT, and instances exist only inside LDS:
//
//   * a tile row is cut into segments of 16 tiles (8 or 4 where the lists are long: segment_shift());
//   * bin_segments (K1): workgroup w takes a contiguous range of Gaussians.  A Gaussian whose tile rectangle covers rows
//     y0..y1 and columns x0..x1 becomes one PAIR per (row, segment) it touches: {key = depth bits << 32 | id, the column
//     range inside the segment as one byte} -- 9 bytes for on average ~1.6 instances.  Two passes over the workgroup's
//     rectangles (they stay in L2): count per segment in LDS, scan, then place every pair with a returning LDS atomic
//     into the workgroup's OWN region of the pair array, ordered by segment.  The region starts at the instance prefix
//     of the workgroup's first Gaussian (from preprocess_fwd's per-block totals; pairs <= instances, so regions never
//     overlap): no global cursor, nothing to clear.  All writes of a region come from one workgroup within microseconds,
//     so its lines leave the L2 complete.  Also per workgroup: the start of each segment's run (pair_off[segment][w]) and
//     the running instance count per segment (inst_pre[segment][w]);
//   * bin_tiles (K2): one 512-thread workgroup per segment.  Its list start is the sum over w of inst_pre[s - 1][w] (the
//     range table needs no device-wide scan).  The segment's pairs -- one short run per K1 workgroup -- are fetched with
//     ALL loads in flight at once (a flat pair index -> source address map is built in LDS first; walking the runs one
//     after the other cost sixteen dependent memory round trips per wave and 56 us), held in registers, counted per tile
//     and placed INTO LDS with LDS atomics; the 16 ranges are written, and every wave sorts whole tile lists in its
//     registers (tile_sort.h: no barrier, no LDS traffic beyond one read) and writes the sorted ids: point_list is written
//     once, coalesced, and keys never exist in global memory.
//     Tile lists above 1024 entries, segments above the LDS budget or with more pairs than the registers hold take a
//     per-tile path (whole workgroup: LDS sort up to 4096 keys, in-place global sort above, in the `keys` scratch array).
//
// HBM traffic per view at config 3: 9 B x 1.0 M pairs written and read once, 4 B x R written -- against 4 + 8 + 4 B written,
// 8 + 4 + 8 + 4 B read per instance plus the 8 MB table three times before.
#include "dgr_common.h"
#include "kernels.h"
#include "tile_sort.h"
#include <mutex>

namespace dgr {
namespace {

constexpr int K1_THREADS = 1024;
constexpr int K2_THREADS = 512;
constexpr int K2_WAVES = K2_THREADS / 64;
constexpr int SEG_MAX = SEG_TILES_MAX;      // tiles per segment: 16, 8 or 4 (chosen per call from the expected list lengths)
constexpr int K2_PPT = 12;                   // pairs a bin_tiles thread holds in registers: 6144 per segment (= K2_CAP; 16 spill)
constexpr int K2_CAP = 6144;        // keys of one segment held in LDS (48 KB: three workgroups per CU)
constexpr int REG_SORT_MAX = 1024;  // a wave sorts a tile list in registers up to here (16 chunks of 64)
constexpr int BIG_PAIRS = 24;       // bin_segments: rectangles of more (row, segment) pairs are walked by a whole wave
constexpr int BIGQ = 512;           // ... from a queue of this many entries per 4096 Gaussians (8 KB of LDS)

// exclusive (inclusive) scan of a[0..n) in place by the whole workgroup; returns the total.  NT threads, all call it.
template <int NT>
__device__ __forceinline__ uint32_t block_scan(uint32_t* a, int n, bool inclusive, uint32_t* wsum, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    const int per = (n + NT - 1) / NT;
    const int lo = min(tid * per, n), hi = min(lo + per, n);
    uint32_t s = 0;
    for (int i = lo; i < hi; i++) s += a[i];
    uint32_t incl = s;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t v = __shfl_up(incl, off, 64);
        if (lane >= off) incl += v;
    }
    __syncthreads();  // (wsum may still be read from a previous call)
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t before = 0, total = 0;
#pragma unroll
    for (int ww = 0; ww < NT / 64; ww++) {
        const uint32_t v = wsum[ww];
        if (ww < wave) before += v;
        total += v;
    }
    uint32_t run = before + incl - s;
    for (int i = lo; i < hi; i++) {
        const uint32_t c = a[i];
        a[i] = inclusive ? run + c : run;
        run += c;
    }
    __syncthreads();
    return total;
}

// ------------------------------------------------------------------------------------------------ K1
// Every thread handles its Gaussians four at a time with the four loads issued together (the kernel is a latency
// skeleton: one memory round trip per group instead of one per Gaussian); a range of at most 4096 Gaussians keeps its
// rectangles and depths in registers between the two passes, longer ranges re-read them (from L2) in pass B.
__global__ void __launch_bounds__(K1_THREADS) bin_segments_kernel(int P, int per_wg, GeometryView geom, SegmentTables tb,
                                                                  uint64_t* __restrict__ pair_keys, uint8_t* __restrict__ pair_cov,
                                                                  int grid_x, int grid_y, int seg_shift, int capacity, int prefixed) {
    // both[nseg]: pairs (low word) and instances (high word) per segment, one 64-bit LDS atomic per pair | cnt[nseg]
    // (pairs; then the start of each segment's run, then the fill cursors) | inst[nseg]
    extern __shared__ unsigned long long lds64[];
    __shared__ uint32_t wsum[K1_THREADS / 64];
    __shared__ uint32_t s_base, s_vis;
    __shared__ uint32_t s_qn;        // the queue of big rectangles (below)
    __shared__ uint4 s_q[BIGQ];      // {x0 | y0 << 16, x1 | y1 << 16, depth bits, Gaussian id}
    const int SEG = 1 << seg_shift;
    const int sgx = (grid_x + SEG - 1) >> seg_shift;
    const int nseg = grid_y * sgx;
    unsigned long long* both = lds64;
    uint32_t* cnt = reinterpret_cast<uint32_t*>(lds64 + nseg);
    uint32_t* inst = cnt + nseg;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wg = blockIdx.x, nwg = gridDim.x;
    const int g0 = wg * per_wg, g1 = min(P, g0 + per_wg);
    constexpr int NH = 4;
    const bool hold = per_wg <= NH * K1_THREADS;
    ushort4 hr[NH];
    float hd[NH];
    auto load_group = [&](int first, bool with_depth) {
#pragma unroll
        for (int k = 0; k < NH; k++) {
            const int idx = first + tid + k * K1_THREADS;
            hr[k] = make_ushort4(0, 0, 0, 0);
            hd[k] = 0.f;
            if (idx < g1) {
                hr[k] = geom.rect[idx];
                if (with_depth) hd[k] = geom.depths[idx];
            }
        }
    };
    if (hold) load_group(g0, true);
    for (int i = tid; i < nseg; i += K1_THREADS) both[i] = 0ull;
    // instances of all Gaussians in front of this workgroup's range = where its region of the pair array starts
    uint32_t part = 0;
    if (prefixed) {
        if (tid == 0) part = geom.block_tiles[g0 >> 8];  // (callback path: scan_blocks left the exclusive prefix)
    } else {
        for (int b = tid; b < (g0 >> 8); b += K1_THREADS) part += geom.block_tiles[b] & 0x7fffffffu;
    }
    __syncthreads();
    // A Gaussian whose rectangle covers many (row, segment) pairs -- the heavy tail of a real map: a splat of 150 px sigma touches
    // 56 rows x 4-8 segments, a full-frame one 68 x 8 at 1080p -- is not walked by the lane that holds it (one lane looping
    // over 500 pairs, an LDS atomic each, beside 63 idle lanes: 149 us for this kernel on dgr_amd.synth.heavy_tail_scene at
    // config 3's size against 16 us on synth-v1) but queued, and the queue is walked by the whole workgroup, a wave per entry,
    // a lane per pair -- the wave-per-rectangle form of the reference's one-thread loop (L/cuda_rasterizer/rasterizer_impl.cu:
    // 94-107).  The queue is rebuilt in each pass (which lane's entry overflows a full queue may differ between the passes: each
    // pass only needs every pair handled once); an entry that finds the queue full is walked by its own lane as before.
    auto pairs_of = [&](ushort4 r) -> int {
        if (r.z <= r.x || r.w <= r.y) return 0;
        return ((int)r.w - (int)r.y) * ((((int)r.z - 1) >> seg_shift) - ((int)r.x >> seg_shift) + 1);
    };
    auto enqueue = [&](ushort4 r, float depth, int idx) -> bool {
        const uint32_t q = atomicAdd(&s_qn, 1u);
        if (q >= (uint32_t)BIGQ) return false;
        s_q[q] = make_uint4((uint32_t)r.x | ((uint32_t)r.y << 16), (uint32_t)r.z | ((uint32_t)r.w << 16), __float_as_uint(depth), (uint32_t)idx);
        return true;
    };
    // every lane of the calling wave: (row, segment) pair number p of entry e -> its segment index and column span
    auto queued_pair = [&](const uint4& e, int p, float inv_nsx, int nsx, int sx0, int& seg, int& xa, int& xb) {
        const int rx = (int)(e.x & 0xffffu), ry = (int)(e.x >> 16), rz = (int)(e.y & 0xffffu);
        int row = (int)(((float)p + 0.5f) * inv_nsx);  // p / nsx for p < 2^22 (p + 1/2 is at least 1/(2 nsx) away from a multiple of nsx)
        const int sx = sx0 + p - row * nsx;
        seg = (ry + row) * sgx + sx;
        xa = max(rx, sx * SEG);
        xb = min(rz, sx * SEG + SEG);
    };
    // ---- pass A: pairs and instances per segment
    auto count = [&](ushort4 r) {
        if (r.z <= r.x || r.w <= r.y) return;
        const int sx0 = r.x >> seg_shift, sx1 = (r.z - 1) >> seg_shift;
        for (int y = r.y; y < r.w; y++)
            for (int sx = sx0; sx <= sx1; sx++)
                atomicAdd(&both[y * sgx + sx],
                          1ull | ((unsigned long long)(min((int)r.z, sx * SEG + SEG) - max((int)r.x, sx * SEG)) << 32));
    };
    uint32_t on_screen = 0;  // this thread's Gaussians with a tile rectangle
    for (int first = g0; first < g1; first += NH * K1_THREADS) {
        if (!hold) load_group(first, false);
        if (tid == 0) s_qn = 0u;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NH; k++) {
            on_screen += (hr[k].z > hr[k].x && hr[k].w > hr[k].y) ? 1u : 0u;
            if (pairs_of(hr[k]) <= BIG_PAIRS || !enqueue(hr[k], 0.f, 0)) count(hr[k]);
        }
        __syncthreads();
        const int nq = min((int)s_qn, BIGQ);
        for (int q = tid >> 6; q < nq; q += K1_THREADS / 64) {
            const uint4 e = s_q[q];
            const int sx0 = (int)(e.x & 0xffffu) >> seg_shift, nsx = ((((int)(e.y & 0xffffu)) - 1) >> seg_shift) - sx0 + 1;
            const int np = ((int)(e.y >> 16) - (int)(e.x >> 16)) * nsx;
            const float inv_nsx = 1.0f / (float)nsx;
            for (int p = lane; p < np; p += 64) {
                int seg, xa, xb;
                queued_pair(e, p, inv_nsx, nsx, sx0, seg, xa, xb);
                atomicAdd(&both[seg], 1ull | ((unsigned long long)(xb - xa) << 32));
            }
        }
        __syncthreads();  // (the queue is rewritten by the next group)
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        part += __shfl_xor(part, off, 64);
        on_screen += __shfl_xor(on_screen, off, 64);
    }
    if (tid == 0) { s_base = 0u; s_vis = 0u; }
    __syncthreads();
    if (lane == 0 && part) atomicAdd(&s_base, part);
    if (lane == 0 && on_screen) atomicAdd(&s_vis, on_screen);
    for (int i = tid; i < nseg; i += K1_THREADS) {
        const unsigned long long v = both[i];
        cnt[i] = (uint32_t)v;
        inst[i] = (uint32_t)(v >> 32);
    }
    __syncthreads();
    const uint32_t base = s_base;
    const uint32_t total_pairs = block_scan<K1_THREADS>(cnt, nseg, false, wsum, tid);
    block_scan<K1_THREADS>(inst, nseg, true, wsum, tid);
    // tables, transposed so that bin_tiles reads rows of nwg consecutive words
    for (int s = tid; s < nseg; s += K1_THREADS) {
        tb.pair_off[(size_t)s * nwg + wg] = base + cnt[s];
        tb.inst_pre[(size_t)s * nwg + wg] = inst[s];
    }
    if (tid == 0) {
        tb.pair_off[(size_t)nseg * nwg + wg] = base + total_pairs;
        tb.pair_off[(size_t)(nseg + 1) * nwg + wg] = s_vis;
    }
    __syncthreads();
    // ---- pass B: place the pairs; cnt[] now serves as the fill cursors
    auto place = [&](ushort4 r, float depth, int idx) {
        if (r.z <= r.x || r.w <= r.y) return;
        const uint64_t key = ((uint64_t)__float_as_uint(depth) << 32) | (uint32_t)idx;
        const int sx0 = r.x >> seg_shift, sx1 = (r.z - 1) >> seg_shift;
        for (int y = r.y; y < r.w; y++)
            for (int sx = sx0; sx <= sx1; sx++) {
                const uint32_t slot = base + atomicAdd(&cnt[y * sgx + sx], 1u);
                const int x0 = max((int)r.x, sx * SEG) - sx * SEG, x1 = min((int)r.z, sx * SEG + SEG) - sx * SEG;  // 0 <= x0 < x1 <= 16
                if (slot < (uint32_t)capacity) {  // (past the capacity bin_tiles flags the overflow and reads nothing)
                    pair_keys[slot] = key;
                    pair_cov[slot] = (uint8_t)(x0 | ((x1 - 1) << 4));
                }
            }
    };
    for (int first = g0; first < g1; first += NH * K1_THREADS) {
        if (!hold) load_group(first, true);
        if (tid == 0) s_qn = 0u;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NH; k++) {
            const int idx = first + tid + k * K1_THREADS;
            if (pairs_of(hr[k]) <= BIG_PAIRS || !enqueue(hr[k], hd[k], idx)) place(hr[k], hd[k], idx);
        }
        __syncthreads();
        const int nq = min((int)s_qn, BIGQ);
        for (int q = tid >> 6; q < nq; q += K1_THREADS / 64) {
            const uint4 e = s_q[q];
            const int sx0 = (int)(e.x & 0xffffu) >> seg_shift, nsx = ((((int)(e.y & 0xffffu)) - 1) >> seg_shift) - sx0 + 1;
            const int np = ((int)(e.y >> 16) - (int)(e.x >> 16)) * nsx;
            const float inv_nsx = 1.0f / (float)nsx;
            const uint64_t key = ((uint64_t)e.z << 32) | e.w;
            for (int p = lane; p < np; p += 64) {
                int seg, xa, xb;
                queued_pair(e, p, inv_nsx, nsx, sx0, seg, xa, xb);
                const uint32_t slot = base + atomicAdd(&cnt[seg], 1u);
                const int x0 = xa & (SEG - 1), x1 = xb - (xa - x0);  // columns inside the segment: 0 <= x0 < x1 <= 16
                if (slot < (uint32_t)capacity) {
                    pair_keys[slot] = key;
                    pair_cov[slot] = (uint8_t)(x0 | ((x1 - 1) << 4));
                }
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------ K2
struct K2Shared {
    union {
        uint64_t keys[K2_CAP];           // the segment's keys, grouped by tile
        uint32_t src[K2_PPT * K2_THREADS];  // before that: flat pair index -> index into the pair array
    };
    uint32_t run_a[SEG_MAX_WGS], run_start[SEG_MAX_WGS + 1];
    uint32_t tcnt[SEG_MAX], tbase[SEG_MAX], tfill[SEG_MAX];
    uint32_t red[4][K2_WAVES];
    uint32_t wsum[K2_WAVES];
    uint32_t counter;
};

// every pair of the segment: f(coverage byte, index into the pair array).  Flat pair index -> run by a binary search in the
// runs' prefix (LDS), four pairs per thread with their loads issued together: a segment too dense for the registers is read in
// n_pairs / 2048 memory round trips.  (Walking the runs one after the other, a wave per run, is one DEPENDENT round trip
// per run: 31 per wave and pass, ~60 us -- and a dense segment makes several passes.)
template <typename F>
__device__ __forceinline__ void for_each_pair(const K2Shared& sh, int nwg, uint32_t n_pairs, const uint8_t* __restrict__ pair_cov,
                                              int tid, F f) {
    for (uint32_t i0 = (uint32_t)tid; i0 < n_pairs; i0 += 4u * K2_THREADS) {
        uint32_t a[4], c[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t i = i0 + (uint32_t)(u * K2_THREADS);
            c[u] = 0xffffffffu;
            a[u] = 0u;
            if (i < n_pairs) {
                int lo = 0, hi = nwg;  // the run holding flat index i: the last r with run_start[r] <= i
                while (hi - lo > 1) {
                    const int mid = (lo + hi) >> 1;
                    if (sh.run_start[mid] <= i) lo = mid; else hi = mid;
                }
                a[u] = sh.run_a[lo] + (i - sh.run_start[lo]);
                c[u] = pair_cov[a[u]];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++)
            if (c[u] != 0xffffffffu) f(c[u], a[u]);
    }
}

// One wave, one tile list of up to REG_SORT_MAX entries (tile_sort.h).  (Round 8 gave frames of long lists -- LONG_LISTS, chosen per
// launch from the expected list length -- a network on the full 64-bit keys, because the 32-bit one needed a four-part rank merge
// above 512 entries; round 9's takes 1024 in one pass and is the faster of the two there as well: bin_tiles 131 -> 82 us at config 4,
// 103 -> 67 on the heavy-tailed scene, profiles/r9/bin_tiles_ab.txt.)  Inlined at its three call sites (236 registers spilled around
// them in all): as a real call -- 45 spilled -- the kernel was a third slower everywhere (synth-v1 25.5 -> 34.3 us, config 4 82 -> 104).
__device__ __forceinline__ void sort_tile_in_wave(uint64_t* src, int n, uint32_t* dst, int lane) {
    if (n <= 64) sort_wave_trunc<1>(src, n, dst, lane);
    else if (n <= 128) sort_wave_trunc<2>(src, n, dst, lane);
    else if (n <= 256) sort_wave_trunc<4>(src, n, dst, lane);
    else if (n <= 512) sort_wave_trunc<8>(src, n, dst, lane);
    else sort_wave_trunc<16>(src, n, dst, lane);
}

// (xcd_contiguous, dgr_common.h: XCD x gets a contiguous run of segments)
__global__ void __launch_bounds__(K2_THREADS, 6) bin_tiles_kernel(ImageView img, uint32_t* __restrict__ point_list,
                                                                  uint64_t* __restrict__ key_scratch, SegmentTables tb,
                                                                  const uint64_t* __restrict__ pair_keys,
                                                                  const uint8_t* __restrict__ pair_cov,
                                                                  const uint32_t* __restrict__ block_tiles, int nblocks, int nwg,
                                                                  int grid_x, int grid_y, int seg_shift, int capacity, int prefixed,
                                                                  int sched_on, StatusReport rep, int xp_map, unsigned long long* trace) {
    __shared__ K2Shared sh;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int SEG = 1 << seg_shift;
    const int sgx = (grid_x + SEG - 1) >> seg_shift;
    const int nseg = grid_y * sgx;
    // debug trace (dgr_debug_bin_tiles_trace): eight words per workgroup, phase time stamps at 100 MHz
    auto stamp = [&](int k) { if (trace && tid == 0) trace[(size_t)blockIdx.x * 8 + k] = wall_clock64(); };
    auto note = [&](int k, unsigned long long v) { if (trace && tid == 0) trace[(size_t)blockIdx.x * 8 + k] = v; };
    stamp(0);
    // Workgroups 0 .. nseg-1 take a segment each.  Behind them come SEG / 4 - 1 HELPERS per segment, one per further group of
    // four tiles: a helper leaves at once unless its segment is DENSE (more keys than LDS or more pairs than the registers
    // hold), in which case the segment's tiles are shared out four to a workgroup instead of being taken in turn by one --
    // on a clustered frame the dense segments are few and everything else has long finished (profiles/r6/clustered.txt).
    const int helpers = (SEG >> 2) - 1;
    const int part = (int)blockIdx.x < nseg ? 0 : 1 + ((int)blockIdx.x - nseg) % max(helpers, 1);
    const int s_lin = (int)blockIdx.x < nseg ? (int)blockIdx.x : ((int)blockIdx.x - nseg) / max(helpers, 1);
    const int s = xp_map ? xcd_contiguous(s_lin, nseg) : s_lin;
    const int ty = s / sgx, sx = s - ty * sgx;
    const int ntl = min(SEG, grid_x - sx * SEG);             // tiles of this segment (the last one of a row may be short)
    const int tile0 = ty * grid_x + sx * SEG;

    // ---- list start, size of the segment, grand total: column sums of the per-workgroup running counts; the runs
    uint32_t before = 0, upto = 0, total = 0, flag = 0;
    for (int w = tid; w < nwg; w += K2_THREADS) {
        if (s > 0) before += tb.inst_pre[(size_t)(s - 1) * nwg + w];
        upto += tb.inst_pre[(size_t)s * nwg + w];
        total += tb.inst_pre[(size_t)(nseg - 1) * nwg + w];
        const uint32_t a = tb.pair_off[(size_t)s * nwg + w];
        sh.run_a[w] = a;
        sh.run_start[w] = tb.pair_off[(size_t)(s + 1) * nwg + w] - a;  // (length; scanned below)
    }
    if (s == 0 && !prefixed)  // the `prefiltered` violation flag of preprocess_fwd's blocks (callback path: scan_blocks took it)
        for (int i = tid; i < nblocks; i += K2_THREADS) flag |= block_tiles[i] >> 31;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        before += __shfl_xor(before, off, 64);
        upto += __shfl_xor(upto, off, 64);
        total += __shfl_xor(total, off, 64);
        flag |= __shfl_xor(flag, off, 64);
    }
    if (lane == 0) { sh.red[0][wave] = before; sh.red[1][wave] = upto; sh.red[2][wave] = total; sh.red[3][wave] = flag; }
    if (tid < SEG_MAX) { sh.tcnt[tid] = 0u; sh.tfill[tid] = 0u; }
    if (tid == 0) sh.run_start[nwg] = 0u;
    __syncthreads();
    before = 0; upto = 0; total = 0; flag = 0;
#pragma unroll
    for (int ww = 0; ww < K2_WAVES; ww++) { before += sh.red[0][ww]; upto += sh.red[1][ww]; total += sh.red[2][ww]; flag |= sh.red[3][ww]; }
    const bool overflow = total > (uint32_t)capacity;
    const uint32_t gcount = upto - before;
    if (s == 0 && part == 0 && wave == 0) {  // (one wave of one workgroup: the frame's status word and blend flags)
        // bit 2 of the flags: the light blend kernels walk one list per QUADRANT wave instead of one per half-wave (render_common.h:
        // blend_slot) -- forced by the caller (BLEND_LISTS_QUADRANT) or, with BLEND_LISTS_AUTO, decided here for THIS frame: where a
        // Gaussian on screen touches more than ten tiles on average, nearly every list entry lives in both halves of its quadrants --
        // nothing for half-wave lists to skip, nothing to pair -- and the finer lists cost more than they save.  The crossing lies
        // between 8.4 and 11.3 tiles per Gaussian on synth-v1 with scaled splats and between 6.3 and 11.9 on the heavy-tailed scene
        // with a growing share of big ones (profiles/r9/lists_sweep.txt; either side of it the two mappings differ by < 1 %).
        bool quadrant_lists = (sched_on & BLEND_LISTS_QUADRANT) != 0;
        if (sched_on & BLEND_LISTS_AUTO) {
            uint32_t vis = 0;  // Gaussians on screen: bin_segments' workgroups left their counts in the tables' last row
            for (int w = lane; w < nwg; w += 64) vis += tb.pair_off[(size_t)(nseg + 1) * nwg + w];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) vis += __shfl_xor(vis, off, 64);
            quadrant_lists = (unsigned long long)total > 10ull * vis;
        }
        if (lane == 0) {
            img.status[0] = (int)total;
            img.status[1] = overflow ? 1 : 0;
            if (!prefixed) {
                img.status[2] = (int)flag;  // prefiltered violation
                img.status[3] = 0;          // full variant: number of valid (pixel, Gaussian) pairs, summed by its forward blend
            }
            img.cursor[2] = (uint32_t)capacity;
            // bit 0: the blend kernels walk tile_sched; bit 1: this frame overflowed
            img.cursor[3] = (uint32_t)(sched_on & 1) | (overflow ? 2u : 0u) | (quadrant_lists ? 4u : 0u);
        }
    }
    if (overflow || gcount == 0u) {  // (empty tiles keep {0, 0}: the reference clears the table and writes only tiles that own instances)
        if (part == 0 && tid < ntl) img.ranges[tile0 + tid] = make_uint2(0u, 0u);
        return;
    }
    const uint32_t n_pairs = block_scan<K2_THREADS>(sh.run_start, nwg + 1, false, sh.wsum, tid);  // run_start[w] = pairs of the runs < w
    stamp(1);
    note(6, (unsigned long long)gcount | ((unsigned long long)n_pairs << 32));
    const bool in_regs = n_pairs <= (uint32_t)(K2_PPT * K2_THREADS);
    const bool dense = !in_regs || gcount > (uint32_t)K2_CAP;
    if (part > 0 && !dense) return;

    // ---- the segment's pairs into registers: flat index -> source map in LDS (one thread per run), then every load at once
    uint64_t pk[K2_PPT];
    uint32_t pc[K2_PPT];
    if (!dense) {
        for (int w = tid; w < nwg; w += K2_THREADS) {
            const uint32_t a = sh.run_a[w], b0 = sh.run_start[w], len = sh.run_start[w + 1] - b0;
            for (uint32_t j = 0; j < len; j++) sh.src[b0 + j] = a + j;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < K2_PPT; k++) {
            const uint32_t i = (uint32_t)(tid + k * K2_THREADS);
            pc[k] = 0xffffffffu;
            pk[k] = 0ull;
            if (i < n_pairs) {
                const uint32_t a = sh.src[i];
                pc[k] = pair_cov[a];
                pk[k] = pair_keys[a];
            }
        }
        // ---- pass A: instances per tile
#pragma unroll
        for (int k = 0; k < K2_PPT; k++)
            if (pc[k] != 0xffffffffu)
                for (uint32_t x = pc[k] & 15u; x <= (pc[k] >> 4); x++) atomicAdd(&sh.tcnt[x], 1u);
    } else {
        for_each_pair(sh, nwg, n_pairs, pair_cov, tid, [&](uint32_t c, uint32_t) {
            for (uint32_t x = c & 15u; x <= (c >> 4); x++) atomicAdd(&sh.tcnt[x], 1u);
        });
    }
    __syncthreads();  // (also: every thread is done with sh.src, which shares its bytes with sh.keys)
    stamp(2);
    if (wave == 0) {
        const uint32_t c = (lane < SEG_MAX) ? sh.tcnt[lane] : 0u;
        uint32_t incl = c;
#pragma unroll
        for (int off = 1; off < SEG_MAX; off <<= 1) {
            const uint32_t v = __shfl_up(incl, off, 64);
            if (lane >= off) incl += v;
        }
        if (lane < SEG_MAX) sh.tbase[lane] = incl - c;
        if (part == 0 && lane < ntl) img.ranges[tile0 + lane] = c ? make_uint2(before + incl - c, before + incl) : make_uint2(0u, 0u);
    }
    __syncthreads();
    // General sorter (dense segments, and segments with a list above REG_SORT_MAX): the lists of tiles [ta, tb_) lie in sh.keys
    // from offset tbase[ta] on.  Lists up to REG_SORT_MAX entries are dealt to the waves round-robin (tile_sort.h: one wave's
    // registers); longer ones are cut into parts of 512, the parts dealt to the waves as well, and every list is then merged
    // by rank with all threads searching.
    constexpr int WAVE_MAX = REG_SORT_MAX;
    auto sort_lists = [&](int ta, int tb_) {
        const uint32_t gb = sh.tbase[ta];
        int w = 0;
        int parts_seen = 0;  // the parts of the longer lists are dealt to the waves as well ...
        for (int t = ta; t < tb_; t++) {
            const int n = (int)sh.tcnt[t];
            if (n == 0) continue;
            uint64_t* list = sh.keys + (sh.tbase[t] - gb);
            if (n <= WAVE_MAX) {
                if ((w++ % K2_WAVES) == wave) sort_tile_in_wave(list, n, point_list + before + sh.tbase[t], lane);
            } else {
                for (int p = 0; p < list_parts(n); p++)
                    if ((w++ % K2_WAVES) == wave) sort_list_part(list, n, p, lane);
                parts_seen++;
            }
        }
        if (parts_seen == 0) return;
        __syncthreads();  // ... and every thread then helps merging them, list after list
        for (int t = ta; t < tb_; t++) {
            const int n = (int)sh.tcnt[t];
            if (n > WAVE_MAX) merge_list_parts<K2_THREADS>(sh.keys + (sh.tbase[t] - gb), n, point_list + before + sh.tbase[t], tid);
        }
    };

    uint32_t tmax = 0;
#pragma unroll
    for (int t = 0; t < SEG_MAX; t++) tmax = max(tmax, sh.tcnt[t]);
    // the frame's longest tile list, for the forward's status report (StatusReport, dgr_common.h; the forward blend's first
    // workgroup delivers it): one atomic without a return value per segment, nothing waits for it.  (A ticket here, with the
    // last workgroup reporting, was built first: in the middle of the kernel it cost wave 0 two memory round trips -- 27 -> 37 us
    // at config 3 -- and as an object whose destructor ran at every return it made the LONG_LISTS instantiation, which spills,
    // fault on a reloaded pointer.)
    if (rep.ws && part == 0 && tid == 0 && tmax) __hip_atomic_fetch_max(rep.ws, tmax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!dense) {
        // ---- pass B: the segment's keys into LDS, grouped by tile
#pragma unroll
        for (int k = 0; k < K2_PPT; k++)
            if (pc[k] != 0xffffffffu)
                for (uint32_t x = pc[k] & 15u; x <= (pc[k] >> 4); x++) sh.keys[sh.tbase[x] + atomicAdd(&sh.tfill[x], 1u)] = pk[k];
        __syncthreads();
        stamp(3);
        note(7, (unsigned long long)tmax | ((unsigned long long)s << 32));
        if (tmax <= (uint32_t)REG_SORT_MAX) {
            // ---- the rule: every wave sorts whole tile lists in its registers and writes the ids
            for (int t = wave; t < ntl; t += K2_WAVES) {
                const int n = (int)sh.tcnt[t];
                if (n > 0) sort_tile_in_wave(sh.keys + sh.tbase[t], n, point_list + before + sh.tbase[t], lane);
            }
            if (trace) { __syncthreads(); stamp(4); }
            return;
        }
        sort_lists(0, ntl);
        if (trace) { __syncthreads(); stamp(4); }
        return;
    }
    note(7, (unsigned long long)tmax | ((unsigned long long)s << 32) | (1ull << 63));
    // ---- dense segment: this workgroup's four tiles (the whole segment where it has no helpers), in consecutive GROUPS
    // whose keys fit LDS together.  Per group ONE pass over the pairs (re-read from the pair array: held in registers across
    // the sorts they would spill) places the keys; a single list above K2_CAP entries is sorted in place in the `keys`
    // scratch array.  (Round 5 took every tile of such a segment on its own, in one workgroup -- a pass over the pairs, run by
    // run, and a padded workgroup network each: 473-514 us for the kernel on a clustered frame.)
    const int tq0 = helpers > 0 ? part * 4 : 0, tq1 = helpers > 0 ? min(ntl, tq0 + 4) : ntl;
    for (int t0 = tq0; t0 < tq1;) {
        const int n0 = (int)sh.tcnt[t0];
        if (n0 == 0) { t0++; continue; }
        __syncthreads();  // the previous group's sorts have read sh.keys
        if (n0 > K2_CAP) {
            uint64_t* gk = key_scratch + before + sh.tbase[t0];
            uint32_t* pl = point_list + before + sh.tbase[t0];
            if (tid == 0) sh.counter = 0u;
            __syncthreads();
            for_each_pair(sh, nwg, n_pairs, pair_cov, tid, [&](uint32_t c, uint32_t a) {
                if ((c & 15u) <= (uint32_t)t0 && (uint32_t)t0 <= (c >> 4)) gk[atomicAdd(&sh.counter, 1u)] = pair_keys[a];
            });
            __syncthreads();
            wg_sort_global<K2_THREADS>(gk, n0, tid);
            for (int i = tid; i < n0; i += K2_THREADS) pl[i] = (uint32_t)gk[i];
            t0++;
            continue;
        }
        int t1 = t0;
        uint32_t sum = 0;
        while (t1 < tq1 && sum + sh.tcnt[t1] <= (uint32_t)K2_CAP) sum += sh.tcnt[t1++];
        const uint32_t gb = sh.tbase[t0];
        for_each_pair(sh, nwg, n_pairs, pair_cov, tid, [&](uint32_t c, uint32_t a) {
            if ((c & 15u) < (uint32_t)t1 && (uint32_t)t0 <= (c >> 4)) {
                const uint32_t xa = max((uint32_t)t0, c & 15u), xb = min((uint32_t)t1 - 1u, c >> 4);
                const uint64_t key = pair_keys[a];
                for (uint32_t x = xa; x <= xb; x++) sh.keys[sh.tbase[x] - gb + atomicAdd(&sh.tfill[x], 1u)] = key;
            }
        });
        __syncthreads();
        stamp(3);
        sort_lists(t0, t1);
        t0 = t1;
    }
    if (trace) { __syncthreads(); stamp(4); }
}

}  // namespace

unsigned long long* g_bin_tiles_trace = nullptr;  // (dgr_debug_bin_tiles_trace)

// the segment tables must fit one workgroup's LDS in bin_segments (16 bytes per segment, at the smallest segment size)
bool segment_binning_fits(int W, int H) {
    const int gx = tiles_x(W), gy = tiles_y(H);
    if (gx <= 0 || gy <= 0) return false;
    const size_t nseg = (size_t)gy * ((gx + SEG_TILES_MIN - 1) / SEG_TILES_MIN);
    return nseg * 16 <= SEG_K1_LDS_MAX;
}
// Tiles per segment, as a shift: the largest of 16, 8, 4 for which an average segment stays inside bin_tiles' LDS budget.
// `capacity` is what the caller expects to render (the lazy bindings size it 1.5x above the largest count seen, the
// callback path passes the exact count): segments are sized so that capacity / tiles * segment <= 1.5 K2_CAP, i.e. an
// average segment fills at most two thirds (lazy) or all (exact) of the budget; denser segments take the per-tile path.
int segment_shift(int W, int H, int capacity, int longest_list) {
    const int gx = tiles_x(W), gy = tiles_y(H);
    const long tiles = (long)gx * gy;
    const long per_tile = tiles > 0 ? ((long)(capacity > 0 ? capacity : 0) + tiles - 1) / tiles : 0;
    static const int forced = [] { const char* e = getenv("DGR_SEG_SHIFT"); return (e && e[0] >= '2' && e[0] <= '4' && e[1] == 0) ? e[0] - '0' : 0; }();
    if (forced) return forced;  // (A/B runs)
    int sh = 2;
    for (int c = 4; c > 2; c--)
        if ((per_tile << c) * 2 <= (long)K2_CAP * 3) { sh = c; break; }
    // small frames: bin_tiles is one workgroup per segment and each is a chain of dependent phases -- 90 workgroups on 256
    // CUs (640x480 at 16 tiles per segment) took as long as 544 (1920x1080); prefer at least two workgroups per CU
    while (sh > 2 && (long)gy * ((gx + (1 << sh) - 1) >> sh) < 512) sh--;
    // The frame's longest tile list, where the last forward of this shape reported it (api.hip: hinted_longest_list): a frame
    // whose lists are uneven -- a cluster of 1000-entry lists in a frame that averages 220 -- has segments that overflow the
    // budget although the average one fits, and each of those takes the dense path (every phase several times slower:
    // profiles/r9/bin_tiles_trace_before.txt).  Smaller segments, as long as one made of such lists still fits; if not even
    // four of them do, the capacity's choice stands (the heavy-tailed scene: 108 us at 8 tiles per segment, 133 at 4).
    if (longest_list > 0)
        for (int c = sh; c >= 2; c--)
            if (((long)longest_list << c) <= (long)K2_CAP) { sh = c; break; }
    return sh;
}
// Gaussians per bin_segments workgroup (a multiple of 1024, so that workgroup ranges start at a 256-block boundary) and the
// number of workgroups: at most SEG_MAX_WGS
int segment_binning_per_wg(int P) {
    const int chunks = (P + K1_THREADS - 1) / K1_THREADS;
    const int rounds = (chunks + SEG_MAX_WGS - 1) / SEG_MAX_WGS;
    return max(1, rounds) * K1_THREADS;
}
int segment_binning_workgroups(int P) { return max(1, (P + segment_binning_per_wg(P) - 1) / segment_binning_per_wg(P)); }

hipError_t launch_bin_segments(int P, GeometryView geom, BinningView bin, SegmentTables tb, int grid_x, int grid_y, int seg_shift,
                               int capacity, bool prefixed, hipStream_t stream) {
    const int nseg = grid_y * ((grid_x + (1 << seg_shift) - 1) >> seg_shift);
    const size_t lds = (size_t)nseg * 16;
    const int per_wg = segment_binning_per_wg(P);
    auto kernel = bin_segments_kernel;
    static std::mutex mu;
    static bool attr_set[64] = {};
    int dev = 0;
    hipError_t rc = hipGetDevice(&dev);
    if (rc != hipSuccess) return rc;
    {   // the attribute is per device (and cheap): set it once for each device this process drives
        std::lock_guard<std::mutex> lk(mu);
        if (dev < 0 || dev >= 64 || !attr_set[dev]) {
            rc = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, SEG_K1_LDS_MAX);
            if (rc != hipSuccess) return rc;
            if (dev >= 0 && dev < 64) attr_set[dev] = true;
        }
    }
    launch_shmem(kernel, dim3(segment_binning_workgroups(P)), dim3(K1_THREADS), lds, stream, P, per_wg, geom, tb, bin.pair_keys,
                 bin.pair_cov, grid_x, grid_y, seg_shift, capacity, prefixed ? 1 : 0);
    return hipGetLastError();
}
hipError_t launch_bin_tiles(int P, GeometryView geom, ImageView img, BinningView bin, SegmentTables tb, int grid_x, int grid_y,
                            int seg_shift, int capacity, bool prefixed, int blend_flags, StatusReport rep, hipStream_t stream) {
    const int nseg = grid_y * ((grid_x + (1 << seg_shift) - 1) >> seg_shift);
    const int helpers = (1 << seg_shift) / 4 - 1;  // per segment (bin_tiles_kernel)
    // block -> segment map: XCD x takes a contiguous run of segments on a frame known to be even (the last report of this shape
    // switched the tile schedule off); otherwise consecutive segments go to consecutive XCDs -- the segments of a cluster are
    // neighbours, and a contiguous map hands all of them to two or three of the eight XCDs (clustered scene: 129 -> 107 us
    // before anything else changed, profiles/r9/bin_tiles_ab.txt).  DGR_BT_MAP = 0 / 1 forces one (A/B runs).
    static const int forced_map = [] { const char* e = getenv("DGR_BT_MAP"); return (e && (e[0] == '0' || e[0] == '1')) ? e[0] - '0' : -1; }();
    const int xp_map = forced_map >= 0 ? forced_map : ((blend_flags & 1) ? 0 : 1);
    launch(bin_tiles_kernel, dim3(nseg * (1 + helpers)), dim3(K2_THREADS), stream, img, bin.point_list, bin.keys, tb, bin.pair_keys, bin.pair_cov,
           geom.block_tiles, (P + 255) / 256, segment_binning_workgroups(P), grid_x, grid_y, seg_shift, capacity, prefixed ? 1 : 0,
           blend_flags, rep, xp_map, g_bin_tiles_trace);
    return hipGetLastError();
}

}  // namespace dgr
