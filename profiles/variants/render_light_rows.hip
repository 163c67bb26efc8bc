// render_light_rows.hip -- backward alpha blending of the light variant (L/cuda_rasterizer/backward.cu:419-699) with one
// 4x4 pixel block per 16-lane DPP row.
//
// render_light.hip's backward gives a wave one 8x8 quadrant and visits every (quadrant, Gaussian) pair some pixel of
// the quadrant blended: 1.41 M pairs per 1080p / 500 k view with 16 of 64 lanes valid on average, ~100 VALU issue slots
// each, 45 of them a 64-lane butterfly -- the kernel is bound by VALU issue.  Here every 16-lane row
// of a wave owns one 4x4 block and walks ITS OWN list of the Gaussians some pixel of that block blended (the
// forward's 16-bit contribution tags): 2.92 M (block, Gaussian) pairs, four of them per wave instruction, 0.84-0.88 M
// wave iterations in all, and the reduction of a pair's twelve sums is a 16-lane DPP network (25 instructions for the
// four pairs of an iteration together) instead of a 64-lane one per pair.
// What made the first attempt at this mapping slower ("one 4x4 sub-block per DPP row") was the merge of
// the blocks' partial sums: twice as many LDS float atomics.  Here a (block, Gaussian) pair owns a private 12-float
// entry that it writes once with a plain store; the entries of one Gaussian are contiguous (entry base per list
// position = running sum of the popcounts of the tags), so the collect phase is a short contiguous sum per Gaussian
// followed by the usual line-coalesced global atomics -- no LDS atomics except the rare median-depth term.
// The sums leave as RAW moments of q = o G dL/dalpha over the pixel offsets; preprocess_bwd applies the conic /
// opacity / ndc factors once per Gaussian (PreprocessBwdArgs::acc_raw).
#include "render_common.h"

#ifndef DGR_ABLATE
#define DGR_ABLATE 0  // 1 / 2: measurement builds (profiles/ablate.sh), never shipped
#endif

namespace dgr {
namespace {

constexpr int RB_NB = 128;  // list positions staged per round
constexpr int RB_E = 384;   // private 12-float entries per round (a round ends early when the tags would need more)
constexpr int RB_EC = 12;
constexpr int RB_LIST_LD = RB_NB + 4;

struct RowsStage {
    float4 rec[2 * (RB_NB + 1)];  // [2 s] = {x, y, a2, c2}, [2 s + 1] = {b2, opacity, -, -}; slot RB_NB: sentinel (opacity 0)
    float4 rgbd[RB_NB + 1];       // {r, g, b, depth}
    uint32_t id[RB_NB];
    uint32_t ebt[RB_NB + 1];      // low 16 bits: first entry of the slot, high 16 bits: its tag
    unsigned char list[16][RB_LIST_LD];  // per 4x4 block: slots whose tag has the block's bit, ascending
    int cnt2[2][16];              // entries of list[b] contributed by staging wave 0 / 1
    float ent[RB_E * RB_EC];
    float med[RB_NB];             // sum of dL/dmedian over the pixels whose median Gaussian the slot is
    int wsum[2], wdrop[2];
    int max_last;
};

// ---- twelve values reduced over the 16 lanes of every row at once (25 DPP instructions) -----------------------------
// Afterwards lane L = lane & 15 of each row holds the row total of value row_reduce12_comp(L); the lanes with
// (L & 3) == 3 duplicate their neighbour.
__device__ __forceinline__ int row_reduce12_comp(int lane) {
    const int b3 = (lane >> 3) & 1, b2 = (lane >> 2) & 1, b1 = (lane >> 1) & 1;
    return (lane & 1) ? 8 + 2 * b2 + b3 : 4 * b1 + 2 * b2 + b3;
}
__device__ __forceinline__ float row_reduce12(const float (&x)[12], int lane) {
    // stage 1: l <-> 15 - l; lanes 0-7 of a row keep x[2 i], lanes 8-15 x[2 i + 1] (bank-masked writes).  One block, the
    // two writes of a destination six instructions apart; the leading s_nop covers the DPP-after-VALU-write hazard.
    float y0, y1, y2, y3, y4, y5;
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %6, %6 row_mirror row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %1, %8, %8 row_mirror row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %2, %10, %10 row_mirror row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %3, %12, %12 row_mirror row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %4, %14, %14 row_mirror row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %5, %16, %16 row_mirror row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %0, %7, %7 row_mirror row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %1, %9, %9 row_mirror row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %2, %11, %11 row_mirror row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %3, %13, %13 row_mirror row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %4, %15, %15 row_mirror row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %5, %17, %17 row_mirror row_mask:0xf bank_mask:0xc\n\t"
        : "=&v"(y0), "=&v"(y1), "=&v"(y2), "=&v"(y3), "=&v"(y4), "=&v"(y5)
        : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]), "v"(x[8]), "v"(x[9]),
          "v"(x[10]), "v"(x[11]));
    // stage 2: l <-> 7 - l inside each half row; lanes 0-3 / 8-11 keep y[2 j], lanes 4-7 / 12-15 y[2 j + 1]
    float z0, z1, z2;
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %3, %3 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %1, %5, %5 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %2, %7, %7 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %0, %4, %4 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %1, %6, %6 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %2, %8, %8 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        : "=&v"(z0), "=&v"(z1), "=&v"(z2)
        : "v"(y0), "v"(y1), "v"(y2), "v"(y3), "v"(y4), "v"(y5));
    // stage 3: l <-> l ^ 2; bit 1 of the lane picks z0 / z1; z2 is summed into both halves
    const bool b1 = (lane & 2) != 0, b0 = (lane & 1) != 0;
    const float s = b1 ? z1 : z0, so = b1 ? z0 : z1;
    const float m01 = s + dpp_mov<DPP_QUAD_XOR2>(so);
    const float m2 = z2 + dpp_mov<DPP_QUAD_XOR2>(z2);
    // stage 4: l <-> l ^ 1; bit 0 picks m01 / m2
    const float t = b0 ? m2 : m01, to = b0 ? m01 : m2;
    return t + dpp_mov<DPP_QUAD_XOR1>(to);
}

__device__ __forceinline__ int popc16(unsigned v) { return __builtin_popcount(v & 0xffffu); }

template <bool DO_POSE>
__global__ void __launch_bounds__(256, 5) render_bwd_light_rows_kernel(RenderBwdLightArgs a) {
    __shared__ RowsStage sb;
    const int tile = xcd_tile(blockIdx.x, a.grid_x * a.grid_y);
    const int tx = tile % a.grid_x, ty = tile / a.grid_x;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, row = lane >> 4;
    const int blk = 4 * wave + row;  // this lane's 4x4 block: quadrant wave, block row = 2 (y / 4) + x / 4 -- as the forward tags them
    const int px = tx * DGR_BLOCK_X + (wave & 1) * 8 + (row & 1) * 4 + (lane & 3);
    const int py = ty * DGR_BLOCK_Y + (wave >> 1) * 8 + (row >> 1) * 4 + ((lane >> 2) & 3);
    const bool inside = px < a.W && py < a.H;
    const size_t pix_id = (size_t)a.W * py + px;
    const size_t N = (size_t)a.W * a.H;
    const f2 pxy = {(float)px, (float)py};

    const uint2 range = a.ranges[tile];
    const int last_contributor = inside ? (int)a.n_contrib[pix_id] : 0;
    const uint16_t* tags16 = carve_binning(const_cast<char*>(a.binning_base), (size_t)*a.capacity).tags16;

    if (tid == 0) {
        sb.max_last = 0;
        sb.rec[2 * RB_NB] = make_float4(0.f, 0.f, 0.f, 0.f);
        sb.rec[2 * RB_NB + 1] = make_float4(0.f, 0.f, 0.f, 0.f);  // opacity 0: never valid
        sb.rgbd[RB_NB] = make_float4(0.f, 0.f, 0.f, 0.f);
        sb.ebt[RB_NB] = 0u;
    }
    __syncthreads();
    {
        int v = last_contributor;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off, 64));
        if (lane == 0) atomicMax(&sb.max_last, v);
    }
    __syncthreads();
    const int total = min((int)(range.y - range.x), sb.max_last);  // nothing past the last contributor matters
    if (total <= 0) return;

    const float T_final = inside ? (1.0f - a.alphas[pix_id]) : 0.f;
    float T = T_final;
    float dpix0 = 0.f, dpix1 = 0.f, dpix2 = 0.f, dpix_depth = 0.f, dpix_median = 0.f, dpix_var = 0.f, gt_px = 0.f;
    if (inside) {
        dpix0 = a.dL_dpix[pix_id];
        dpix1 = a.dL_dpix[N + pix_id];
        dpix2 = a.dL_dpix[2 * N + pix_id];
        dpix_depth = a.dL_dpix_depth[pix_id];
        dpix_median = a.dL_dpix_median[pix_id];
        dpix_var = a.dL_dpix_var[pix_id];
        gt_px = a.gt_depth[pix_id];
    }
    const float bg_term = -T_final * (a.bg[0] * dpix0 + a.bg[1] * dpix1 + a.bg[2] * dpix2);
    const float dvar2 = 2.f * dpix_var;
    // one scalar recurrence instead of the reference's five (render_light.hip): S <- alpha X + (1 - alpha) S
    float S = 0.f, X_last = 0.f, last_alpha = 0.f, last_om = 1.f;
    bool mid_once = true;
    const int L = lane & 15;
    const int my_comp = ((L & 3) == 3) ? -1 : row_reduce12_comp(L);  // value this lane delivers after the row reduction
    const unsigned below = (1u << blk) - 1u;

    for (int hi = total; hi > 0;) {
        const int lo0 = max(0, hi - RB_NB), win = hi - lo0;
        __syncthreads();  // previous round collected
        // ---- which positions of the window fit this round: suffix sums of the tags' popcounts (positions are consumed
        // back to front, so the top of the window always belongs to the round)
        uint32_t entry = 0u;
        unsigned tg = 0u;
        if (tid < win) {
            entry = a.point_list[range.x + lo0 + tid];
            tg = tags16[range.x + lo0 + tid];
        }
        int c = popc16(tg), suf = c;  // inclusive suffix sum inside the wave
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int v = __shfl_down(suf, off, 64);
            if (lane + off < 64) suf += v;
        }
        if (wave < 2 && lane == 0) sb.wsum[wave] = suf;
        if (tid < RB_NB) sb.med[tid] = 0.f;
        __syncthreads();
        if (wave == 0) suf += sb.wsum[1];
        const bool keep = tid < win && suf <= RB_E;  // monotone in tid; the top position always fits (<= 16 entries)
        const int ebase = suf - c;                   // entries of the positions above this one
        // the sixteen lists (slots ascending) come from ballots of the two staging waves
        unsigned long long bal[16];
#pragma unroll
        for (int b = 0; b < 16; b++) {
            bal[b] = __ballot(keep && ((tg >> b) & 1u));
            if (lane == 0 && wave < 2) sb.cnt2[wave][b] = __popcll(bal[b]);
        }
        {   // cut = number of window positions left for the next round
            const unsigned long long db = __ballot(tid < win && !keep);
            if (lane == 0 && wave < 2) sb.wdrop[wave] = __popcll(db);
        }
        __syncthreads();
        const int cut = sb.wdrop[0] + sb.wdrop[1];
        const int lo = lo0 + cut, cnt = hi - lo;
        const int slot = tid - cut;
        // ---- stage the kept, tagged positions
        if (keep) {
            sb.ebt[slot] = (uint32_t)ebase | (tg << 16);
            if (tg) {
                const uint32_t gid = entry & ID_MASK;
                const float4 q0 = a.rec[3 * (size_t)gid + 0];
                const float4 q1 = a.rec[3 * (size_t)gid + 1];
                const float4 q2 = a.rec[3 * (size_t)gid + 2];
                sb.rec[2 * slot] = make_float4(q0.x, q0.y, -0.5f * PSCALE * q1.x, -0.5f * PSCALE * q1.z);
                sb.rec[2 * slot + 1] = make_float4(-PSCALE * q1.y, q0.w, 0.f, 0.f);
                sb.rgbd[slot] = make_float4(q2.x, q2.y, q2.z, q0.z);
                sb.id[slot] = gid;
            }
            if (tg) {
#pragma unroll
                for (int b = 0; b < 16; b++)
                    if ((tg >> b) & 1u) sb.list[b][(wave ? sb.cnt2[0][b] : 0) + lanes_below(bal[b])] = (unsigned char)slot;
            }
        }
        __syncthreads();

        // ---- every row walks its block's list back to front
        const int n_row = sb.cnt2[0][blk] + sb.cnt2[1][blk];
        int n_it = n_row;
        n_it = max(n_it, __shfl_xor(n_it, 16, 64));
        n_it = max(n_it, __shfl_xor(n_it, 32, 64));
        n_it = __builtin_amdgcn_readfirstlane(n_it);
#if DGR_ABLATE == 1
        n_it *= (a.W < 0 ? 1 : 0);  // (measurement build: no pair loop, nothing else removed)
#endif
        const int rel_last = last_contributor - lo;  // slots below this are at or before the last contributor
        const unsigned char* my_list = sb.list[blk];
        for (int it = 0; it < n_it; it++) {
            const int k = n_row - 1 - it;
            const int j = (k >= 0) ? (int)my_list[k] : RB_NB;
            const float4 q0 = sb.rec[2 * j], q1 = sb.rec[2 * j + 1];
            const float4 cd = sb.rgbd[j];
            const uint32_t et = sb.ebt[j];
            f2 dxy;
            const float p2 = pair_p2(q0, q1, pxy, dxy);
            const float dx = dxy.x, dy = dxy.y;
            const float oG = alpha_raw(q1.y, p2);
            const float alpha = fminf(0.99f, oG);
            const bool valid = (j < rel_last) & (p2 <= 0.0f) & (alpha >= ALPHA_MIN);
            float w = 0.f, qq = 0.f, e = 0.f;
            if (valid) {
                const float om = 1.f - alpha;
                const float inv = recip(om);
                T = T * inv;
                w = alpha * T;
                e = cd.w - gt_px;
                const float X = cd.x * dpix0 + cd.y * dpix1 + cd.z * dpix2 + cd.w * dpix_depth + (e * e) * dpix_var;
                S = last_alpha * X_last + last_om * S;
                X_last = X;
                last_alpha = alpha;
                last_om = om;
                const float dL_dalpha = (X - S) * T + bg_term * inv;
                qq = oG * dL_dalpha;
                if (T > 0.5f && mid_once) {  // backward.cu:654-664, once per pixel; the per-Gaussian factors are applied later
                    atomicAdd(&sb.med[j], dpix_median);
                    mid_once = false;
                }
            }
            const float qdx = qq * dx, qdy = qq * dy;
            const float wd = w * dpix_depth;
            float g[12];
            g[0] = w * dpix0;
            g[1] = w * dpix1;
            g[2] = w * dpix2;
            g[3] = wd + (dvar2 * w) * e;
            g[4] = qdx;       // Sx
            g[5] = qdy;       // Sy
            g[6] = qdx * dx;  // Sxx
            g[7] = qdx * dy;  // Sxy
            g[8] = qdy * dy;  // Syy
            g[9] = qq;        // S0
            g[10] = DO_POSE ? wd : 0.f;
            g[11] = 0.f;
#if DGR_ABLATE == 2
            const float tot = ((g[0] + g[1]) + (g[2] + g[3])) + ((g[4] + g[5]) + (g[6] + g[7])) + ((g[8] + g[9]) + g[10]);  // (no row reduction)
#else
            const float tot = row_reduce12(g, lane);
#endif
            if (my_comp >= 0 && k >= 0)
                sb.ent[((et & 0xffffu) + (unsigned)popc16((et >> 16) & below)) * RB_EC + my_comp] = tot;
        }
        __syncthreads();

        // ---- collect: a slot's entries are contiguous; 16 lanes per slot, lane = component
        {
            const int comp = tid & 15;
            if (comp < 12) {
                for (int s = tid >> 4; s < cnt; s += 16) {
                    const uint32_t et = sb.ebt[s];
                    const int ne = popc16(et >> 16);
                    if (ne == 0) continue;
                    const float* p = sb.ent + (et & 0xffffu) * RB_EC + comp;
                    float v = 0.f;
                    for (int i = 0; i < ne; i++) v += p[i * RB_EC];
                    if (comp == 11) v = sb.med[s];  // (value 11 of the reduction is unused: the lane carries the median sum)
                    // accumulator components (dgr_common.h): 0..9 as reduced, 10 = median sum, 13 = pose depth sum
                    const int dst = comp == 10 ? 13 : comp == 11 ? 10 : comp;
                    if (v != 0.f) atomicAdd(a.acc + (size_t)sb.id[s] * DGR_ACC_STRIDE + dst, v);
                }
            }
        }
        hi = lo;
    }
}

__global__ void __launch_bounds__(64) row_reduce_test_kernel(const float* in, float* out, int* comp) {
    const int lane = threadIdx.x;
    float g[12];
#pragma unroll
    for (int k = 0; k < 12; k++) g[k] = in[k * 64 + lane];
    out[lane] = row_reduce12(g, lane);
    comp[lane] = ((lane & 3) == 3) ? -1 : row_reduce12_comp(lane & 15);
}

}  // namespace

hipError_t launch_render_bwd_light_rows(const RenderBwdLightArgs& a, hipStream_t stream) {
    const int tiles = a.grid_x * a.grid_y;
    if (tiles <= 0) return hipSuccess;
    if (!a.track_off)
        launch((render_bwd_light_rows_kernel<true>), dim3(tiles), dim3(256), stream, a);
    else
        launch((render_bwd_light_rows_kernel<false>), dim3(tiles), dim3(256), stream, a);
    return hipGetLastError();
}
hipError_t launch_row_reduce_test(const float* in, float* out, int* comp, hipStream_t stream) {
    launch(row_reduce_test_kernel, dim3(1), dim3(64), stream, in, out, comp);
    return hipGetLastError();
}

}  // namespace dgr
