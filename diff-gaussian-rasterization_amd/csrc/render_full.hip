// render_full.hip -- blend kernels of the -full variant for gfx950.
//
// Replaces renderCUDA forward (F/cuda_rasterizer/forward.cu:261-396), renderCUDA backward
// (F/cuda_rasterizer/backward.cu:540-836) and the per-pixel half of ComputePG (:838-1338).
// Traversal, staging and reductions are those of render_light.hip (render_common.h); what differs:
//  forward : the terminating Gaussian IS blended, then the pixel stops (forward.cu:370-381); outputs colour,
//            depth and "uncertainty" = sum alpha T; keeps final T, n_contrib, the number of valid contributors
//            (their total is the reference's num_related_primitives) and the position of the first one;
//  backward: T_final comes from the stored final T; the uncertainty channel is differentiated as the variance
//            sum (d - gt)^2 alpha T (backward.cu:701-708; the forward/backward mismatch is the fork's, quirk F2).
//  pose    : the reference stores 92 bytes per valid (pixel, Gaussian) pair in NG-sized lists and re-walks every
//            tile in ComputePG.  Its result is linear in per-Gaussian sums: part 1 needs sum w dL_dpixel (= the
//            colour gradient), part 2-1 the colour-only dL/d(ndc) sums, and the depth terms -- assigned, not
//            accumulated, there (:1278-1289) -- touch only each pixel's front-most valid Gaussian, which the
//            forward recorded.  Five more accumulator components replace the lists, the second tile walk and
//            the second host sync; part 2-2 is dead code in the reference (:1264-1275) and is not computed.
#include "render_common.h"

namespace dgr {
namespace {

// ================================================================================ forward
__global__ void __launch_bounds__(256) render_fwd_full_kernel(RenderFwdFullArgs a) {
    __shared__ Staged s;
    __shared__ int s_nvalid;
    const int tile = xcd_tile(blockIdx.x, a.grid_x * a.grid_y);
    const int tx = tile % a.grid_x, ty = tile / a.grid_x;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int px = tx * DGR_BLOCK_X + (wave & 1) * 8 + (lane & 7);
    const int py = ty * DGR_BLOCK_Y + (wave >> 1) * 8 + (lane >> 3);
    const bool inside = px < a.W && py < a.H;
    const size_t pix_id = (size_t)a.W * py + px;
    const float pxf = (float)px, pyf = (float)py;
    const float tile_x0 = (float)(tx * DGR_BLOCK_X), tile_y0 = (float)(ty * DGR_BLOCK_Y);

    const uint2 range = a.ranges[tile];
    const int total = (int)(range.y - range.x);

    float T = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f, U = 0.f, Dd = 0.f;
    uint32_t last_contributor = 0, first_contributor = 0, nvalid = 0;
    float thr = inside ? ALPHA_MIN : __builtin_inff();
    if (tid == 0) {
        s.rec[2 * SENTINEL] = make_float4(0.f, 0.f, 0.f, 0.f);
        s.rec[2 * SENTINEL + 1] = make_float4(0.f, 0.f, __int_as_float(SENTINEL), 0.f);
        s_nvalid = 0;
    }

    for (int base = 0; base < total; base += DGR_TILE_PIX) {
        if (__syncthreads_and(thr > 1.0f)) break;
        const int cnt = min(DGR_TILE_PIX, total - base);
        unsigned code = 0;
        if (tid < cnt) code = stage_one(s, tid, a.point_list[range.x + base + tid], a.rec, tile_x0, tile_y0, nullptr);
        const int n = build_lists(s, code, tid, wave, lane);

        for (int k = 0; k < n; k += 4) {
            float4 q0[4], q1[4];
            load4(s, wave, k, q0, q1);
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const float dx = q0[u].x - pxf, dy = q0[u].y - pyf;
                const float p2 = dx * (q0[u].z * dx + q0[u].w * dy) + q1[u].x * dy * dy;
                const float alpha = fminf(0.99f, q1[u].y * __builtin_amdgcn_exp2f(p2));
                if (p2 <= 0.0f && alpha >= thr) {
                    const int j = __float_as_int(q1[u].z);
                    const float4 cd = s.rgbd[j];
                    const float w = alpha * T;
                    C0 += cd.x * w; C1 += cd.y * w; C2 += cd.z * w;
                    Dd += cd.w * w;
                    U += w;
                    nvalid++;
                    T = T * (1.0f - alpha);
                    last_contributor = (uint32_t)(base + j + 1);
                    if (first_contributor == 0) first_contributor = last_contributor;
                    if (T < 0.0001f) thr = __builtin_inff();  // blended first, then done (forward.cu:370-381)
                }
            }
            if (__all(thr > 1.0f)) break;
        }
    }

    if (inside) {
        const size_t N = (size_t)a.W * a.H;
        a.final_T[pix_id] = T;
        a.n_valid[pix_id] = nvalid;
        a.n_contrib[pix_id] = last_contributor;
        a.first_contrib[pix_id] = first_contributor;
        a.out_color[pix_id] = C0 + T * a.bg[0];
        a.out_color[N + pix_id] = C1 + T * a.bg[1];
        a.out_color[2 * N + pix_id] = C2 + T * a.bg[2];
        a.out_depth[pix_id] = Dd;
        a.out_uncertainty[pix_id] = U;
    }
    // num_related_primitives = total of n_valid_contrib (F/cuda_rasterizer/rasterizer_impl.cu:495-498)
    {
        int v = inside ? (int)nvalid : 0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        __syncthreads();
        if (lane == 0) atomicAdd(&s_nvalid, v);
        __syncthreads();
        if (tid == 0 && s_nvalid) atomicAdd(&a.status[3], s_nvalid);
    }
}

// ================================================================================ backward
constexpr int NACC_FULL = 15;

struct StagedBwdFull {
    Staged f;
    float4 raw[DGR_TILE_PIX];  // {conic a, b, c, unused}
    float acc[NACC_FULL * ACC_LD];
    int max_last;
};

__global__ void __launch_bounds__(256) render_bwd_full_kernel(RenderBwdFullArgs a) {
    __shared__ StagedBwdFull sb;
    Staged& s = sb.f;
    const int tile = xcd_tile(blockIdx.x, a.grid_x * a.grid_y);
    const int tx = tile % a.grid_x, ty = tile / a.grid_x;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int px = tx * DGR_BLOCK_X + (wave & 1) * 8 + (lane & 7);
    const int py = ty * DGR_BLOCK_Y + (wave >> 1) * 8 + (lane >> 3);
    const bool inside = px < a.W && py < a.H;
    const size_t pix_id = (size_t)a.W * py + px;
    const size_t N = (size_t)a.W * a.H;
    const float pxf = (float)px, pyf = (float)py;
    const float tile_x0 = (float)(tx * DGR_BLOCK_X), tile_y0 = (float)(ty * DGR_BLOCK_Y);

    const uint2 range = a.ranges[tile];
    const int last_contributor = inside ? (int)a.n_contrib[pix_id] : 0;
    const int first_contributor = inside ? (int)a.first_contrib[pix_id] : 0;

    if (tid == 0) {
        sb.max_last = 0;
        s.rec[2 * SENTINEL] = make_float4(0.f, 0.f, 0.f, 0.f);
        s.rec[2 * SENTINEL + 1] = make_float4(0.f, 0.f, __int_as_float(SENTINEL), 0.f);
    }
    __syncthreads();
    {
        int v = last_contributor;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off, 64));
        if (lane == 0) atomicMax(&sb.max_last, v);
    }
    __syncthreads();
    const int total = min((int)(range.y - range.x), sb.max_last);
    if (total <= 0) return;

    const float T_final = inside ? a.final_T[pix_id] : 0.f;  // backward.cu:598
    float T = T_final;
    float dpix0 = 0.f, dpix1 = 0.f, dpix2 = 0.f, dL_depth = 0.f, dL_dunc = 0.f, gt_px = 0.f;
    if (inside) {
        dpix0 = a.dL_dpix[pix_id];
        dpix1 = a.dL_dpix[N + pix_id];
        dpix2 = a.dL_dpix[2 * N + pix_id];
        dL_depth = a.dL_depths[pix_id];
        dL_dunc = a.dL_duncertainties[pix_id];
        gt_px = a.gt_depth[pix_id];
    }
    const float bg_dot_dpixel = a.bg[0] * dpix0 + a.bg[1] * dpix1 + a.bg[2] * dpix2;
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc_depth = 0.f, acc_unc = 0.f;
    float last_alpha = 0.f, lc0 = 0.f, lc1 = 0.f, lc2 = 0.f, last_depth = 0.f, last_unc = 0.f;
    const float ddelx_dx = 0.5f * a.W, ddely_dy = 0.5f * a.H;
    const int c16 = wave_reduce16_comp(lane);
    const int my_comp = ((lane & 3) == 0 && c16 < NACC_FULL) ? c16 : -1;

    for (int hi = total; hi > 0; hi -= DGR_TILE_PIX) {
        const int lo = max(0, hi - DGR_TILE_PIX);
        const int cnt = hi - lo;
        __syncthreads();
        unsigned code = 0;
        if (tid < cnt) code = stage_one(s, tid, a.point_list[range.x + lo + tid], a.rec, tile_x0, tile_y0, &sb.raw[tid]);
#pragma unroll
        for (int k = 0; k < NACC_FULL; k++) sb.acc[k * ACC_LD + tid] = 0.f;
        const int n = build_lists(s, code, tid, wave, lane);
        const int rel_last = last_contributor - lo;
        const int rel_first = first_contributor - 1 - lo;  // slot of the front-most valid contributor, if in this batch

        for (int k = ((n + 3) & ~3) - 4; k >= 0; k -= 4) {
            float4 q0[4], q1[4];
            load4(s, wave, k, q0, q1);
#pragma unroll
            for (int u = 3; u >= 0; u--) {
                const float dx = q0[u].x - pxf, dy = q0[u].y - pyf;
                const float p2 = dx * (q0[u].z * dx + q0[u].w * dy) + q1[u].x * dy * dy;
                const float G = __builtin_amdgcn_exp2f(p2);
                const float alpha = fminf(0.99f, q1[u].y * G);
                const int j = __float_as_int(q1[u].z);
                const bool valid = j < rel_last && p2 <= 0.0f && alpha >= ALPHA_MIN;
                if (!__any(valid)) continue;

                float g[16];
#pragma unroll
                for (int c = 0; c < 16; c++) g[c] = 0.f;
                if (valid) {
                    const float4 cd = s.rgbd[j];
                    const float4 rc = sb.raw[j];
                    const float opac = q1[u].y;
                    const float inv = __builtin_amdgcn_rcpf(1.f - alpha);
                    T = T * inv;
                    const float w = alpha * T;  // dchannel_dcolor
                    const float om = 1.f - last_alpha;
                    acc0 = last_alpha * lc0 + om * acc0; lc0 = cd.x;
                    acc1 = last_alpha * lc1 + om * acc1; lc1 = cd.y;
                    acc2 = last_alpha * lc2 + om * acc2; lc2 = cd.z;
                    const float dcol = (cd.x - acc0) * dpix0 + (cd.y - acc1) * dpix1 + (cd.z - acc2) * dpix2;
                    float dL_dalpha = dcol;
                    const float c_d = cd.w;
                    const float e = c_d - gt_px;
                    const float c_u = e * e;
                    acc_depth = last_alpha * last_depth + om * acc_depth; last_depth = c_d;
                    acc_unc = last_alpha * last_unc + om * acc_unc; last_unc = c_u;
                    dL_dalpha += (c_d - acc_depth) * dL_depth;
                    dL_dalpha += (c_u - acc_unc) * dL_dunc;
                    const float ddepth_dalpha = T * (c_d - acc_depth);  // backward.cu:709
                    dL_dalpha *= T;
                    last_alpha = alpha;
                    dL_dalpha += (-T_final * inv) * bg_dot_dpixel;

                    const float dL_dG = opac * dL_dalpha;
                    const float gdx = G * dx, gdy = G * dy;
                    const float dG_ddelx = -gdx * rc.x - gdy * rc.y;
                    const float dG_ddely = -gdy * rc.z - gdx * rc.y;
                    const float sx = opac * dG_ddelx * ddelx_dx, sy = opac * dG_ddely * ddely_dy;
                    g[0] = w * dpix0;
                    g[1] = w * dpix1;
                    g[2] = w * dpix2;
                    g[3] = w * dL_depth + 2.f * e * w * dL_dunc;  // backward.cu:708
                    g[4] = dL_dG * dG_ddelx * ddelx_dx;
                    g[5] = dL_dG * dG_ddely * ddely_dy;
                    g[6] = -0.5f * gdx * dx * dL_dG;
                    g[7] = -0.5f * gdx * dy * dL_dG;
                    g[8] = -0.5f * gdy * dy * dL_dG;
                    g[9] = G * dL_dalpha;
                    // pose, part 2-1: sum_ch dL_dpixel[ch] * dpixel_dalpha[ch] = T * dcol (colour terms only)
                    const float dla_col = T * dcol;
                    g[10] = dla_col * sx;
                    g[11] = dla_col * sy;
                    if (j == rel_first) {  // the pair ComputePG matches last: its dd_dvK survive (backward.cu:1278-1289)
                        g[12] = dL_depth * w;
                        g[13] = dL_depth * (ddepth_dalpha * sx);
                        g[14] = dL_depth * (ddepth_dalpha * sy);
                    }
                }
                const float tot = wave_reduce16(g, lane);
                if (my_comp >= 0) atomicAdd(&sb.acc[my_comp * ACC_LD + j], tot);
            }
        }
        __syncthreads();
        flush_acc<NACC_FULL>(sb.acc, s.id, cnt, a.acc, tid);
    }
}

}  // namespace

hipError_t launch_render_fwd_full(const RenderFwdFullArgs& a, hipStream_t stream) {
    const int tiles = a.grid_x * a.grid_y;
    if (tiles <= 0) return hipSuccess;
    hipLaunchKernelGGL(render_fwd_full_kernel, dim3(tiles), dim3(256), 0, stream, a);
    return hipGetLastError();
}
hipError_t launch_render_bwd_full(const RenderBwdFullArgs& a, hipStream_t stream) {
    const int tiles = a.grid_x * a.grid_y;
    if (tiles <= 0) return hipSuccess;
    hipLaunchKernelGGL(render_bwd_full_kernel, dim3(tiles), dim3(256), 0, stream, a);
    return hipGetLastError();
}

}  // namespace dgr
