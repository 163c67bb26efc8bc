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
// contribution tags of the batch staged at list position pos0 -> the entries' tag bytes, per half of a quadrant (render_common.h:
// tag_byte; render_light.hip: flush_slot).  Every staged entry gets its byte, blended or not: the bytes underneath are the binning's.
template <class S>
__device__ __forceinline__ void flush_tags(const S& s, const uint32_t* hit, uint8_t* tag8, uint32_t pos0, int tid, bool staged) {
    if (staged) tag8[pos0 + tid] = (uint8_t)tag_byte(hit[tid], __float_as_uint(s.rec[2 * tid + 1].z));
}

// Lists: one per HALF of a quadrant, a loop step serves both half-waves (render_common.h: build_half_lists; render_light.hip)
template <int AM>
__global__ void __launch_bounds__(256, 8) render_fwd_full_kernel(RenderFwdFullArgs a) {
    __shared__ StagedT<DGR_TILE_PIX, unsigned short, 8> s;
    __shared__ uint32_t hit[DGR_TILE_PIX];  // byte w of word j: the UPPER half (lanes 0-31) of quadrant wave w blended staged instance j
                                            // (the lower halves mark byte w of the staged record's spare third word: render_light.hip)
    __shared__ int s_nvalid;
    __shared__ uint64_t exptab[32];         // ALPHA_GLIBC: exact_math.h
    if (a.rep.host && blockIdx.x == 0 && threadIdx.x == 0) report_status(a.rep, a.status);
    bool overflowed;
    const uint4 slot = blend_slot(a.sched, a.ranges, a.sched_flag, a.grid_x * a.grid_y, &overflowed);  // {tile, list start, list end}
    const int tile = (int)slot.x;
    const int tx = tile % a.grid_x, ty = tile / a.grid_x;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int px = tx * DGR_BLOCK_X + (wave & 1) * 8 + (lane & 7);
    const int py = ty * DGR_BLOCK_Y + (wave >> 1) * 8 + (lane >> 3);
    const bool inside = px < a.W && py < a.H;
    const size_t pix_id = (size_t)a.W * py + px;
    const f2 pxy = {(float)px, (float)py};
    const float tile_x0 = (float)(tx * DGR_BLOCK_X), tile_y0 = (float)(ty * DGR_BLOCK_Y);
    const int my_list = 2 * wave + (lane >> 5);
    uint8_t* const tag8 = half_tags(a.point_list, a.sched_flag);
    unsigned char* const mark_base = lane >= 32 ? reinterpret_cast<unsigned char*>(&s.rec[1].z) + wave : reinterpret_cast<unsigned char*>(hit) + wave;
    const int mark_stride = lane >= 32 ? 32 : 4;

    const uint2 range = make_uint2(slot.y, slot.z);
    const int total = (int)(range.y - range.x);

    float T = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f, U = 0.f, Dd = 0.f;
    uint32_t last_contributor = 0, first_contributor = 0, nvalid = 0;
    float ub = inside ? 0.f : -__builtin_inff();  // finished pixels accept nothing (see render_light.hip)
    if (tid == 0) {
        write_sentinel(s);
        s_nvalid = 0;
    }
    if (AlphaPath<AM>::TABLE) exp_ref_table_fill(exptab, tid);  // (visible after the first batch's barriers)

    bool have_flush = false;
    int last_base = 0;
    for (int base = 0; base < total; base += DGR_TILE_PIX) {
        if (__syncthreads_and(ub < 0.f)) break;
        if (have_flush) flush_tags(s, hit, tag8, range.x + base - DGR_TILE_PIX, tid, true);  // (an earlier batch is always full)
        hit[tid] = 0u;
        have_flush = true;
        last_base = base;
        const int cnt = min(DGR_TILE_PIX, total - base);
        unsigned code = 0;
        if (tid < cnt) code = stage_one<AM, true>(s, tid, a.point_list[range.x + base + tid], a.rec, tile_x0, tile_y0);
        const int n = build_half_lists(s, code, tid, wave, lane);

        for (int k = 0; k < n; k += 2) {
            float4 q0[2], q1[2];
            load2(s, my_list, k, q0, q1);
#pragma unroll
            for (int u = 0; u < 2; u++) {
                f2 dxy;
                const float p2 = pair_p2<AM>(q0[u], q1[u], pxy, dxy);
                if ((p2 <= ub) & (p2 >= q1[u].w)) {
                  const float alpha = fminf(0.99f, alpha_raw<AM>(q1[u].y, p2, exptab));
                  if (alpha >= ALPHA_MIN) {
#pragma clang fp contract(off)  // T (1 - alpha) and the sum of alpha T round as the reference's do (forward.cu:366-381)
                    const int j = __float_as_int(q1[u].w) & 0xFF;  // (stage_one<AM, true>: the slot rides in the threshold's low bits)
                    const float4 cd = s.rgbd[j];
                    mark_base[j * mark_stride] = 1;  // contribution tag of this lane's half
                    const float w = alpha * T;
                    C0 = __builtin_fmaf(cd.x, w, C0); C1 = __builtin_fmaf(cd.y, w, C1); C2 = __builtin_fmaf(cd.z, w, C2);
                    Dd = __builtin_fmaf(cd.w, w, Dd);
                    U = U + w;
                    nvalid++;
                    T = T * (1.0f - alpha);
                    last_contributor = (uint32_t)(base + j + 1);
                    if (first_contributor == 0) first_contributor = last_contributor;
                    if (T < 0.0001f) ub = -__builtin_inff();  // blended first, then done (forward.cu:370-381)
                  }
                }
            }
            if (!wave_any(ub >= 0.f)) break;
        }
    }

    __syncthreads();
    if (have_flush) flush_tags(s, hit, tag8, range.x + last_base, tid, tid < total - last_base);
    // the tail of a list whose tile finished early was never staged: nobody blended it (render_common.h: the tag bytes' invariant)
    for (int p = (have_flush ? last_base + DGR_TILE_PIX : 0) + tid; p < total; p += DGR_TILE_PIX) tag8[range.x + p] = 0;

    // (an overflowed forward rendered empty lists: NaN images instead of a plausible empty frame -- render_light.hip)
    if (overflowed) {
        C0 = C1 = C2 = U = Dd = __builtin_nanf("");
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

// list positions staged per batch (occupancy: see render_light.hip).  DET (dgr_set_option("deterministic_grads", 1); round 9 for this
// variant): one accumulator plane per quadrant wave, plain stores, the planes added in wave order, the finished row of a
// (tile, Gaussian) pair STORED to its own row of an instance-major buffer that det_gather_kernel adds up per Gaussian in ascending
// order -- render_light.hip has the scheme; four planes are four times the accumulators, hence 64 positions per batch.
template <bool DET>
struct StagedBwdFull {
    static constexpr int NB = DET ? 64 : 128;
    static constexpr int LD = NB + 1;
    static constexpr int PLANE = NACC_FULL * LD;
    typedef StagedT<NB, uint32_t, DET ? 4 : 8> staged_t;  // (paired lists: two list rows per quadrant wave, render_common.h)
    staged_t f;
    float acc[(DET ? 4 : 1) * PLANE];
    uint32_t inst[DET ? NB : 1];
    int max_last;
    uint64_t exptab[32];  // ALPHA_GLIBC: exact_math.h
};

// Paired lists (round 9, as the light mapping backward: render_light.hip, render_common.h: build_paired_lists): from the forward's
// tags per half, neighbouring entries of a quadrant's list that live in different halves share a loop step, whose sixteen sums are
// reduced per half (wave_reduce16d_head + quad sums) and delivered to each half's own entry.  Not in the deterministic kernel
// (its planes take the LDS).
// LEAN (round 9, as in the light variant): the caller passed no gradient image for the "uncertainty" output (NULL: the loss did not
// use it) -- the variance recurrence and its two terms drop out: bit-identical to the kernel fed an all-zero image.
template <int AM, bool DET = false, bool LEAN = false>
__global__ void __launch_bounds__(256, 6) render_bwd_full_kernel(RenderBwdFullArgs a) {
    typedef StagedBwdFull<DET> SB;
    constexpr int BWD_NB = SB::NB, BWD_LD = SB::LD;
    constexpr bool PAIRED = !DET;
    __shared__ SB sb;
    typename SB::staged_t& s = sb.f;
    const uint4 slot = blend_slot(a.sched, a.ranges, a.sched_flag, a.grid_x * a.grid_y);  // {tile, list start, list end}
    const int tile = (int)slot.x;
    const int tx = tile % a.grid_x, ty = tile / a.grid_x;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int px = tx * DGR_BLOCK_X + (wave & 1) * 8 + (lane & 7);
    const int py = ty * DGR_BLOCK_Y + (wave >> 1) * 8 + (lane >> 3);
    const bool inside = px < a.W && py < a.H;
    const size_t pix_id = (size_t)a.W * py + px;
    const size_t N = (size_t)a.W * a.H;
    const f2 pxy = {(float)px, (float)py};

    const uint2 range = make_uint2(slot.y, slot.z);
    const int last_contributor = inside ? (int)a.n_contrib[pix_id] : 0;
    const int first_contributor = inside ? (int)a.first_contrib[pix_id] : 0;

    if (tid == 0) {
        sb.max_last = 0;
        write_sentinel<true>(s);
    }
    if (AlphaPath<AM>::TABLE) exp_ref_table_fill(sb.exptab, tid);
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
        if (!LEAN && a.dL_duncertainties) dL_dunc = a.dL_duncertainties[pix_id];
        gt_px = a.gt_depth[pix_id];
    }
    const float bg_term = -T_final * (a.bg[0] * dpix0 + a.bg[1] * dpix1 + a.bg[2] * dpix2);  // times 1/(1 - alpha): background term of dL/dalpha
    const float dunc2 = 2.f * dL_dunc;
    // Linear recurrences instead of the reference's five accum_rec_* (see render_light.hip): the full variant needs
    // the colour part and the depth part of dL/dalpha separately (pose terms), hence three scalars:
    //   Xc = <rgb_j, dL/dpixel>, Xd = depth_j, Xu = (depth_j - gt)^2 ; S* <- alpha X* + (1 - alpha) S* after the pair used S*
    float Sc = 0.f, Sd = 0.f, Su = 0.f;
    const float ddelx_dx = 0.5f * a.W, ddely_dy = 0.5f * a.H;
    const int c16 = wave_reduce16d_comp(lane);
    const int my_comp = ((lane & 3) == 0 && c16 < NACC_FULL) ? c16 : -1;
    const uint8_t* const tag8 = half_tags(a.point_list, a.sched_flag);
    // this lane's accumulator row (column = slot); DET: in its wave's own plane
    float* const my_acc = sb.acc + (DET ? wave * SB::PLANE : 0) + (my_comp >= 0 ? my_comp : 0) * BWD_LD;

    for (int hi = total; hi > 0; hi -= BWD_NB) {
        const int lo = max(0, hi - BWD_NB);
        const int cnt = hi - lo;
        __syncthreads();
        unsigned code = 0;
        if (tid < cnt) code = stage_tagged<AM, PAIRED ? TAGS_BYTES_HALVES : TAGS_BYTES_QUADRANT>(s, tid, a.point_list[range.x + lo + tid], a.rec, tag8 + (range.x + lo + tid));
        if (!DET) {  // (DET: a plane's column is written by its wave iff the entry's tag names the wave -- nothing to clear)
#pragma unroll
            for (int k = 0; k < NACC_FULL; k++)
                if (tid < BWD_NB) sb.acc[k * BWD_LD + tid] = 0.f;
        }
        unsigned long long split[2] = {0ull, 0ull};  // PAIRED: the steps that serve two entries
        const int n = PAIRED ? build_paired_lists(s, code, tid, wave, lane, split) : build_lists(s, code, tid, wave, lane);
        // (the staged record carries 4 * slot: render_common.h, stage_tagged)
        const int rel_last4 = 4 * (last_contributor - lo);
        const int rel_first4 = 4 * (first_contributor - 1 - lo);  // 4 * slot of the front-most valid contributor, if in this batch

        // PAIRED: one step per iteration, its list row by half-wave (render_light.hip: the mapping backward)
        constexpr int U = PAIRED ? 1 : 2;
        const int my_list = PAIRED ? 2 * wave + (lane >> 5) : wave;
        for (int k = ((n + U - 1) / U) * U - U; k >= 0; k -= U) {
            float4 q0[2], q1[2];
            if (U == 2) {
                load2(s, my_list, k, q0, q1);
            } else {
                const unsigned off = s.list[my_list][k];
                q0[0] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(s.rec) + off);
                q1[0] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(s.rec) + off + 16);
            }
#pragma unroll
            for (int u = U - 1; u >= 0; u--) {
                f2 dxy;
                const float p2 = pair_p2<AM>(q0[u], q1[u], pxy, dxy);
                const int j4 = __float_as_int(q1[u].z);
                // every listed entry was blended by some pixel of this wave (contribution tags): no wave-level tests
                const float oG = alpha_raw<AM, true>(q1[u].y, p2, sb.exptab);  // o G: alpha before the 0.99 clamp
                // (0.99 is above the threshold, so o G itself decides; "not below" keeps a NaN o G valid as min(0.99f, NaN) does)
                const bool valid = (j4 < rel_last4) & (p2 <= 0.0f) & !(oG < ALPHA_MIN);
                // No branch (see render_light.hip): a lane the Gaussian does not reach runs the same instructions with
                // alpha = 0 and o G = 0 -- 1 / (1 - 0) is exactly 1 and S = 0 X + 1 S keeps its bits, so its state is
                // untouched and all of its contributions are 0; valid lanes execute the reference's operations unchanged.
                // Per-lane scalars: w = alpha T, qq = o G dL/dalpha, qc = o G * (colour-only part of dL/dalpha), and the
                // front-most-pair depth terms fw, fq.
                const float oGm = valid ? oG : 0.f;
                const float am = fminf(0.99f, oGm);  // (one select: render_light.hip)
                const float4 cd = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(s.rgbd) + __float_as_int(q1[u].w));
                const float om = 1.f - am;
                float inv;
                T = t_div<AM>(T, om, inv);  // backward.cu:663
                const float w = am * T;  // dchannel_dcolor
                const float e = cd.w - gt_px;
                const float Xc = cd.x * dpix0 + cd.y * dpix1 + cd.z * dpix2, Xd = cd.w, Xu = e * e;
                const float dcol = Xc - Sc;
                const float ddep = Xd - Sd;
                float dL_dalpha = LEAN ? dcol + ddep * dL_depth : dcol + ddep * dL_depth + (Xu - Su) * dL_dunc;
                dL_dalpha *= T;
                dL_dalpha += bg_term * inv;
                // what the NEXT valid pair (towards the front) subtracts: S <- alpha X + (1 - alpha) S
                Sc = am * Xc + om * Sc;
                Sd = am * Xd + om * Sd;
                if (!LEAN) Su = am * Xu + om * Su;
                const float qq = oGm * dL_dalpha;
                const float qc = oGm * (T * dcol);  // sum_ch dL_dpixel[ch] * dpixel_dalpha[ch] = T * dcol (backward.cu:693)
                // the pair ComputePG matches last: its dd_dvK survive (backward.cu:1278-1289)
                const bool front = valid & (j4 == rel_first4);
                const float fw = front ? dL_depth * w : 0.f;
                const float fq = front ? oG * (dL_depth * (T * ddep)) : 0.f;  // dL_depth * ddepth_dalpha * o * G
                const float dx = dxy.x, dy = dxy.y;
                const float qdx = qq * dx, qdy = qq * dy;
                float g[16];
                g[0] = w * dpix0;
                g[1] = w * dpix1;
                g[2] = w * dpix2;
                g[3] = LEAN ? w * dL_depth : w * dL_depth + (dunc2 * w) * e;  // backward.cu:708
                g[4] = qdx;
                g[5] = qdy;
                g[6] = qdx * dx;
                g[7] = qdx * dy;
                g[8] = qdy * dy;
                g[9] = qq;
                g[10] = qc * dx;  // colour-only sums for pose part 2-1
                g[11] = qc * dy;
                g[12] = fw;
                g[13] = fq * dx;  // front-most depth sums
                g[14] = fq * dy;
                g[15] = 0.f;
                float tot;
                if (PAIRED) {
                    float u0, u1;
                    wave_reduce16d_head(g, u0, u1);
                    const int st = k + u;  // (wave-uniform: scalar code)
                    if ((split[st >> 6] >> (st & 63)) & 1ull) {
                        // a pair: each half's own totals to its own entry's column (j4 is uniform in each half)
                        const float r0 = quad_sum(u0), r1 = quad_sum(u1);
                        const int c0 = (lane & 3) == 0 ? wave_reduce16d_half_slot0(lane) : -1;
                        const int c1 = ((lane & 3) == 0 && wave_reduce16d_half_slot1(lane) < NACC_FULL) ? wave_reduce16d_half_slot1(lane) : -1;
                        char* const col = reinterpret_cast<char*>(sb.acc) + j4;
                        if (c0 >= 0) atomicAdd(reinterpret_cast<float*>(col + c0 * (BWD_LD * 4)), r0);
                        if (c1 >= 0) atomicAdd(reinterpret_cast<float*>(col + c1 * (BWD_LD * 4)), r1);
                        continue;
                    }
                    tot = wave_reduce16d_tail(u0, u1);
                } else {
                    tot = wave_reduce16d(g);  // (within-row stages first: wave_reduce.h)
                }
                if (my_comp >= 0) {
                    float* const cell = reinterpret_cast<float*>(reinterpret_cast<char*>(my_acc) + j4);
                    if (DET) *cell = tot; else atomicAdd(cell, tot);
                }
            }
        }
        __syncthreads();
        if (DET && tid < BWD_NB) {
            // the four planes in wave order into plane 0 (a wave whose tag bit is clear never wrote its column), and the pair's row
            uint32_t row = ~0u;
            if (code != 0u) {
#pragma unroll
                for (int k = 0; k < NACC_FULL; k++) {
                    float v = 0.f;
#pragma unroll
                    for (int w = 0; w < 4; w++)
                        if ((code >> w) & 1u) v += sb.acc[w * SB::PLANE + k * BWD_LD + tid];
                    sb.acc[k * BWD_LD + tid] = v;
                }
                const uint32_t gid = s.id[tid];
                const ushort4 rc = a.det_rect[gid];
                row = a.det_goff[gid] + (uint32_t)(ty - (int)rc.y) * (uint32_t)(rc.z - rc.x) + (uint32_t)(tx - (int)rc.x);
            }
            sb.inst[tid] = row;
        }
        if (DET) __syncthreads();
        // moments -> gradients per staged Gaussian: every "d/d(ndc)" sum is -(a Sx + b Sy) W/2, -(c Sy + b Sx) H/2
        if (code != 0u) {
            constexpr float UN = AlphaPath<AM>::PUNSCALE;  // (undoes the scale of the staged conic)
            const float4 r0 = s.rec[2 * tid], r1 = s.rec[2 * tid + 1];
            const float ca = r0.z * (-2.f * UN), cb = r1.x * (-UN), cc = r0.w * (-2.f * UN);
#pragma unroll
            for (int p = 0; p < 3; p++) {
                const int cx = (p == 0) ? 4 : (p == 1) ? 10 : 13, cy = cx + 1;
                const float Sx = sb.acc[cx * BWD_LD + tid], Sy = sb.acc[cy * BWD_LD + tid];
                sb.acc[cx * BWD_LD + tid] = -(ca * Sx + cb * Sy) * ddelx_dx;
                sb.acc[cy * BWD_LD + tid] = -(cc * Sy + cb * Sx) * ddely_dy;
            }
            sb.acc[6 * BWD_LD + tid] *= -0.5f;
            sb.acc[7 * BWD_LD + tid] *= -0.5f;
            sb.acc[8 * BWD_LD + tid] *= -0.5f;
            sb.acc[9 * BWD_LD + tid] *= __builtin_amdgcn_rcpf(r1.y);
        }
        __syncthreads();
        if (DET) {  // 16 consecutive lanes store one pair's 64-byte row (component 15 stays zero)
            const int comp = tid & 15;
            for (int r = tid >> 4; r < cnt; r += 16) {
                const uint32_t row = sb.inst[r];
                if (row < a.det_R && comp < NACC_FULL) a.det_rows[(size_t)row * DGR_ACC_STRIDE + comp] = sb.acc[comp * BWD_LD + r];
            }
        } else {
            flush_acc<NACC_FULL, BWD_LD>(sb.acc, s.id, cnt, a.acc, tid);
        }
    }
}

}  // namespace

hipError_t launch_render_fwd_full(const RenderFwdFullArgs& a, int alpha_mode, hipStream_t stream) {
    const int tiles = a.grid_x * a.grid_y;
    if (tiles <= 0) return hipSuccess;
    switch (alpha_mode) {
        case ALPHA_FAST: launch_blend(render_fwd_full_kernel<ALPHA_FAST>, dim3(tiles), dim3(256), stream, a); break;
        case ALPHA_GLIBC: launch_blend(render_fwd_full_kernel<ALPHA_GLIBC>, dim3(tiles), dim3(256), stream, a); break;
        default: launch_blend(render_fwd_full_kernel<ALPHA_REF>, dim3(tiles), dim3(256), stream, a);
    }
    return hipGetLastError();
}
hipError_t launch_render_bwd_full(const RenderBwdFullArgs& a, int alpha_mode, hipStream_t stream, bool deterministic) {
    const int tiles = a.grid_x * a.grid_y;
    if (tiles <= 0) return hipSuccess;
    const bool lean = a.dL_duncertainties == nullptr;
    if (deterministic) {  // (alpha_mode 0 only: api.hip)
        if (lean) launch_blend(render_bwd_full_kernel<ALPHA_REF, true, true>, dim3(tiles), dim3(256), stream, a);
        else launch_blend(render_bwd_full_kernel<ALPHA_REF, true>, dim3(tiles), dim3(256), stream, a);
        return hipGetLastError();
    }
    switch (alpha_mode) {
        case ALPHA_FAST: launch_blend(render_bwd_full_kernel<ALPHA_FAST>, dim3(tiles), dim3(256), stream, a); break;   // (reads NULL as zero below)
        case ALPHA_GLIBC: launch_blend(render_bwd_full_kernel<ALPHA_GLIBC>, dim3(tiles), dim3(256), stream, a); break;
        default:
            if (lean) launch_blend(render_bwd_full_kernel<ALPHA_REF, false, true>, dim3(tiles), dim3(256), stream, a);
            else launch_blend(render_bwd_full_kernel<ALPHA_REF>, dim3(tiles), dim3(256), stream, a);
    }
    return hipGetLastError();
}

}  // namespace dgr
