// preprocess.hip -- per-Gaussian kernels for gfx950: forward preprocess, fused backward
// preprocess (+ pose-gradient reduction) and the frustum mark.
//
// Replaces, on the hot path:
//   forward : preprocessCUDA + computeCov3D + computeCov2D + computeColorFromSH
//             (*/cuda_rasterizer/forward.cu:20-256) and the per-tile histogram that stands in for
//             tiles_touched + InclusiveSum (L/cuda_rasterizer/rasterizer_impl.cu:283)
//   backward: computeCov2DCUDA + preprocessCUDA + computeColorFromSH + computeCov3D
//             (L/cuda_rasterizer/backward.cu:20-416) and pose_gradient_preCUDA (:701-751)
//   checkFrustum (L/cuda_rasterizer/rasterizer_impl.cu:54-66)
//
// Both kernels are HBM-bound (one thread per Gaussian, ~250-500 B of traffic each), so the
// arithmetic is written in the reference's association order with FMA contraction OFF: radii,
// tile rects and depth bits -- the integer path -- then agree bit for bit with the CPU oracle.
#include "dgr_common.h"
#include <algorithm>

#include "kernels.h"
#include "count_rank.h"

#pragma clang fp contract(off)
#ifndef DGR_BWD_BATCH_WAVES
#define DGR_BWD_BATCH_WAVES 2  // waves per SIMD the batched backward is compiled for (48 dL_dsh sums live across its view loop)
#endif

namespace dgr {
namespace {

// column-major 3x3: m[c][r]; (A*B)[c][r] = A[0][r]B[c][0] + A[1][r]B[c][1] + A[2][r]B[c][2]
struct M3 {
    float m[3][3];
};
__device__ __forceinline__ M3 mul(const M3& A, const M3& B) {
    M3 R;
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int r = 0; r < 3; r++) R.m[c][r] = A.m[0][r] * B.m[c][0] + A.m[1][r] * B.m[c][1] + A.m[2][r] * B.m[c][2];
    return R;
}
__device__ __forceinline__ M3 transpose(const M3& A) {
    M3 R;
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int r = 0; r < 3; r++) R.m[c][r] = A.m[r][c];
    return R;
}
__device__ __forceinline__ float dot3(float3 a, float3 b) {
    float tx = a.x * b.x, ty = a.y * b.y, tz = a.z * b.z;
    return tx + ty + tz;
}

__device__ __forceinline__ float3 xform4x3(float3 p, const float* m) {
    return make_float3(m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12], m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
                       m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14]);
}
__device__ __forceinline__ float4 xform4x4(float3 p, const float* m) {
    return make_float4(m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12], m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
                       m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14], m[3] * p.x + m[7] * p.y + m[11] * p.z + m[15]);
}

__device__ __forceinline__ float ndc2pix(float v, int S) { return (float)((((double)v + 1.0) * (double)S - 1.0) * 0.5); }

__device__ __forceinline__ void get_rect(float px, float py, int r, int gx, int gy, int& x0, int& y0, int& x1, int& y1) {
    x0 = min(gx, max(0, (int)((px - (float)r) / (float)DGR_BLOCK_X)));
    y0 = min(gy, max(0, (int)((py - (float)r) / (float)DGR_BLOCK_Y)));
    x1 = min(gx, max(0, (int)((px + (float)r + (float)DGR_BLOCK_X - 1.0f) / (float)DGR_BLOCK_X)));
    y1 = min(gy, max(0, (int)((py + (float)r + (float)DGR_BLOCK_Y - 1.0f) / (float)DGR_BLOCK_Y)));
}

constexpr float SH_C0 = 0.28209479177387814f;
constexpr float SH_C1 = 0.4886025119029199f;
__device__ const float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                   -1.0925484305920792f, 0.5462742152960396f};
__device__ const float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                   0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                                   -0.5900435899266435f};

__device__ __forceinline__ float3 operator+(float3 a, float3 b) { return make_float3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ float3 operator-(float3 a, float3 b) { return make_float3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ float3 operator*(float s, float3 a) { return make_float3(s * a.x, s * a.y, s * a.z); }
__device__ __forceinline__ float3 operator*(float3 a, float s) { return make_float3(a.x * s, a.y * s, a.z * s); }

// Loads the SH coefficients a degree needs into registers.  M == 16 rows are 192 B = 12 x 16 B, so a
// lane fetches whole 16-B pieces; consecutive lanes cover one contiguous 12 KB span per wave.
struct SHCoeffs {
    float3 c[16];
};
__device__ __forceinline__ void load_sh(const float* __restrict__ shs, int idx, int deg, int M, bool vec_ok, SHCoeffs& s) {
    const int ncoef = (deg + 1) * (deg + 1);
    if (vec_ok && M == 16) {
        const float4* p = reinterpret_cast<const float4*>(shs + (size_t)idx * 48);
        float f[48];
        const int nvec = (3 * ncoef + 3) >> 2;
#pragma unroll
        for (int i = 0; i < 12; i++) {
            float4 v = (i < nvec) ? p[i] : make_float4(0, 0, 0, 0);
            f[4 * i] = v.x; f[4 * i + 1] = v.y; f[4 * i + 2] = v.z; f[4 * i + 3] = v.w;
        }
#pragma unroll
        for (int k = 0; k < 16; k++) s.c[k] = make_float3(f[3 * k], f[3 * k + 1], f[3 * k + 2]);
    } else {
        const float* p = shs + (size_t)idx * M * 3;
#pragma unroll
        for (int k = 0; k < 16; k++)
            s.c[k] = (k < ncoef && k < M) ? make_float3(p[3 * k], p[3 * k + 1], p[3 * k + 2]) : make_float3(0, 0, 0);
    }
}

// */cuda_rasterizer/forward.cu:74-113 up to `cov` (before the +0.3 low-pass), shared with the backward.
struct Cov2D {
    float3 t;
    float txtz, tytz;
    M3 W, T, Vrk, cov;
};
__device__ __forceinline__ void cov2d_common(float3 mean, float fx, float fy, float tanx, float tany, const float* c3,
                                             const float* v, Cov2D& o) {
    float3 t = xform4x3(mean, v);
    const float limx = 1.3f * tanx, limy = 1.3f * tany;
    o.txtz = t.x / t.z;
    o.tytz = t.y / t.z;
    t.x = fminf(limx, fmaxf(-limx, o.txtz)) * t.z;
    t.y = fminf(limy, fmaxf(-limy, o.tytz)) * t.z;
    o.t = t;
    M3 J;
    J.m[0][0] = fx / t.z; J.m[0][1] = 0.0f; J.m[0][2] = -(fx * t.x) / (t.z * t.z);
    J.m[1][0] = 0.0f; J.m[1][1] = fy / t.z; J.m[1][2] = -(fy * t.y) / (t.z * t.z);
    J.m[2][0] = 0.0f; J.m[2][1] = 0.0f; J.m[2][2] = 0.0f;
    o.W.m[0][0] = v[0]; o.W.m[0][1] = v[4]; o.W.m[0][2] = v[8];
    o.W.m[1][0] = v[1]; o.W.m[1][1] = v[5]; o.W.m[1][2] = v[9];
    o.W.m[2][0] = v[2]; o.W.m[2][1] = v[6]; o.W.m[2][2] = v[10];
    o.T = mul(o.W, J);
    o.Vrk.m[0][0] = c3[0]; o.Vrk.m[0][1] = c3[1]; o.Vrk.m[0][2] = c3[2];
    o.Vrk.m[1][0] = c3[1]; o.Vrk.m[1][1] = c3[3]; o.Vrk.m[1][2] = c3[4];
    o.Vrk.m[2][0] = c3[2]; o.Vrk.m[2][1] = c3[4]; o.Vrk.m[2][2] = c3[5];
    o.cov = mul(mul(transpose(o.T), transpose(o.Vrk)), o.T);
}

__device__ __forceinline__ void quat_to_R(const float4 q, M3& R) {
    const float r = q.x, x = q.y, y = q.z, z = q.w;  // not normalised (forward.cu:127)
    R.m[0][0] = 1.f - 2.f * (y * y + z * z); R.m[0][1] = 2.f * (x * y - r * z); R.m[0][2] = 2.f * (x * z + r * y);
    R.m[1][0] = 2.f * (x * y + r * z); R.m[1][1] = 1.f - 2.f * (x * x + z * z); R.m[1][2] = 2.f * (y * z - r * x);
    R.m[2][0] = 2.f * (x * z - r * y); R.m[2][1] = 2.f * (y * z + r * x); R.m[2][2] = 1.f - 2.f * (x * x + y * y);
}
__device__ __forceinline__ M3 diag3(float a, float b, float c) {
    M3 S;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) S.m[i][j] = 0.0f;
    S.m[0][0] = a; S.m[1][1] = b; S.m[2][2] = c;
    return S;
}

// computeCov3D (forward.cu:118-152): M = S*R, Sigma = M^T M
__device__ __forceinline__ void compute_cov3d(const float* __restrict__ scales, const float* __restrict__ rotations, float mod, int idx,
                                              float (&c3)[6]) {
    const float3 sc = make_float3(scales[3 * idx], scales[3 * idx + 1], scales[3 * idx + 2]);
    const float4 q = make_float4(rotations[4 * idx], rotations[4 * idx + 1], rotations[4 * idx + 2], rotations[4 * idx + 3]);
    M3 R;
    quat_to_R(q, R);
    const M3 S = diag3(mod * sc.x, mod * sc.y, mod * sc.z);
    const M3 Mm = mul(S, R);
    const M3 Sigma = mul(transpose(Mm), Mm);
    c3[0] = Sigma.m[0][0]; c3[1] = Sigma.m[0][1]; c3[2] = Sigma.m[0][2];
    c3[3] = Sigma.m[1][1]; c3[4] = Sigma.m[1][2]; c3[5] = Sigma.m[2][2];
}

}  // namespace

// ---- SH rows <-> lanes through LDS
// A lane that reads (or writes) its own 192-byte SH row touches 12 cache lines, one per instruction, and a wave
// instruction 64 different lines: four times the requests the data needs.  Here a wave moves the rows of 32 Gaussians
// at a time with contiguous 1-KB accesses and transposes them in LDS (row stride 52 dwords: b128-aligned, and the 64
// lanes' rows fall on all banks evenly).  Used for blocks that lie entirely inside [0, P).
constexpr int SHT_ROWS = 32, SHT_LD = 52;
// Every wave transposes through ITS OWN slab of the buffer, so the stores and the loads that follow them need no
// workgroup barrier: the LDS executes one wave's instructions in order; the fence keeps the compiler from reordering.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
// A 16-byte store that is not kept in the caches: the dL_dsh rows (96 MB per view) are written once and read by nobody here.
// Left dirty in L2 / the memory-side cache they were written back under the NEXT kernel's reads: preprocess_fwd of the following
// view 45 -> 39.5 us, the view 0.513 -> 0.508 ms one at a time (profiles/r6/ab_nontemporal.txt).
__device__ __forceinline__ void store_streaming(float4* dst, float4 v) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    __builtin_nontemporal_store(v4f{v.x, v.y, v.z, v.w}, reinterpret_cast<v4f*>(dst));
}
// ... and a 16-byte load likewise: the SH rows (96 MB per view, read once by preprocess_fwd) no longer push the render records
// the same kernel writes out of the caches in front of the forward blend's gathers: one view at a time 0.510 -> 0.501 ms
// (preprocess_fwd -1 us, bin_tiles -0.7, render_fwd -2.5), several views in flight unchanged.
__device__ __forceinline__ float4 load_streaming(const float4* src) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    const v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(src));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void sh_rows_to_lanes(const float* __restrict__ src, size_t g_block, float* lds, float (&f)[48]) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float* my = lds + wave * (SHT_ROWS * SHT_LD);
    const float4* base = reinterpret_cast<const float4*>(src) + (g_block + (size_t)wave * 64) * 12;
#pragma unroll
    for (int r = 0; r < 2; r++) {
        const float4* p = base + (size_t)r * SHT_ROWS * 12;
#pragma unroll
        for (int i = 0; i < 6; i++) {
            const int e = i * 64 + lane;  // float4 index inside the 32-row slab
            const int g = e / 12, c = e - 12 * g;
            *reinterpret_cast<float4*>(my + g * SHT_LD + 4 * c) = load_streaming(p + e);
        }
        wave_sync();
        if ((lane >> 5) == r) {
            const float* row = my + (lane & 31) * SHT_LD;
#pragma unroll
            for (int c = 0; c < 12; c++) {
                const float4 v = *reinterpret_cast<const float4*>(row + 4 * c);
                f[4 * c] = v.x; f[4 * c + 1] = v.y; f[4 * c + 2] = v.z; f[4 * c + 3] = v.w;
            }
        }
        wave_sync();
    }
}
__device__ __forceinline__ void lanes_to_sh_rows(const float (&f)[48], float* __restrict__ dst, size_t g_block, float* lds) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float* my = lds + wave * (SHT_ROWS * SHT_LD);
    float4* base = reinterpret_cast<float4*>(dst) + (g_block + (size_t)wave * 64) * 12;
#pragma unroll
    for (int r = 0; r < 2; r++) {
        if ((lane >> 5) == r) {
            float* row = my + (lane & 31) * SHT_LD;
#pragma unroll
            for (int c = 0; c < 12; c++)
                *reinterpret_cast<float4*>(row + 4 * c) = make_float4(f[4 * c], f[4 * c + 1], f[4 * c + 2], f[4 * c + 3]);
        }
        wave_sync();
        float4* p = base + (size_t)r * SHT_ROWS * 12;
#pragma unroll
        for (int i = 0; i < 6; i++) {
            const int e = i * 64 + lane;
            const int g = e / 12, c = e - 12 * g;
            store_streaming(p + e, *reinterpret_cast<const float4*>(my + g * SHT_LD + 4 * c));
        }
        wave_sync();
    }
}

// The same for rows of the form coef[k] * rgb (dL_dsh: every coefficient's gradient is a scalar times the colour
// gradient): the 48 products are formed while the row is written, so that only 16 + 3 registers stay live.
__device__ __forceinline__ void lanes_to_sh_rows_scaled(const float (&coef)[16], float3 rgb, float* __restrict__ dst, size_t g_block,
                                                        float* lds) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float* my = lds + wave * (SHT_ROWS * SHT_LD);
    float4* base = reinterpret_cast<float4*>(dst) + (g_block + (size_t)wave * 64) * 12;
    const float ch[3] = {rgb.x, rgb.y, rgb.z};
#pragma unroll
    for (int r = 0; r < 2; r++) {
        if ((lane >> 5) == r) {
            float* row = my + (lane & 31) * SHT_LD;
#pragma unroll
            for (int c = 0; c < 12; c++)  // element e = 4 c + j of the row is coefficient e / 3, channel e % 3
                *reinterpret_cast<float4*>(row + 4 * c) =
                    make_float4(coef[(4 * c) / 3] * ch[(4 * c) % 3], coef[(4 * c + 1) / 3] * ch[(4 * c + 1) % 3],
                                coef[(4 * c + 2) / 3] * ch[(4 * c + 2) % 3], coef[(4 * c + 3) / 3] * ch[(4 * c + 3) % 3]);
        }
        wave_sync();
        float4* p = base + (size_t)r * SHT_ROWS * 12;
#pragma unroll
        for (int i = 0; i < 6; i++) {
            const int e = i * 64 + lane;
            const int g = e / 12, c = e - 12 * g;
            store_streaming(p + e, *reinterpret_cast<const float4*>(my + g * SHT_LD + 4 * c));
        }
        wave_sync();
    }
}

// d(colour)/d(unit view direction) of computeColorFromSH, term by term as the reference's backward writes it
// (L/cuda_rasterizer/backward.cu:50-133)
__device__ __forceinline__ void sh_direction_derivatives(const SHCoeffs& s, int D, float3 dir, float3& dRGBdx, float3& dRGBdy,
                                                         float3& dRGBdz) {
    const float x = dir.x, y = dir.y, z = dir.z;
    if (D > 0) {
        dRGBdx = -SH_C1 * s.c[3];
        dRGBdy = -SH_C1 * s.c[1];
        dRGBdz = SH_C1 * s.c[2];
        if (D > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            dRGBdx = dRGBdx + (SH_C2[0] * y * s.c[4] + SH_C2[2] * 2.f * -x * s.c[6] + SH_C2[3] * z * s.c[7] + SH_C2[4] * 2.f * x * s.c[8]);
            dRGBdy = dRGBdy + (SH_C2[0] * x * s.c[4] + SH_C2[1] * z * s.c[5] + SH_C2[2] * 2.f * -y * s.c[6] + SH_C2[4] * 2.f * -y * s.c[8]);
            dRGBdz = dRGBdz + (SH_C2[1] * y * s.c[5] + SH_C2[2] * 2.f * 2.f * z * s.c[6] + SH_C2[3] * x * s.c[7]);
            if (D > 2) {
                dRGBdx = dRGBdx + (SH_C3[0] * s.c[9] * 3.f * 2.f * xy + SH_C3[1] * s.c[10] * yz + SH_C3[2] * s.c[11] * -2.f * xy +
                                   SH_C3[3] * s.c[12] * -3.f * 2.f * xz + SH_C3[4] * s.c[13] * (-3.f * xx + 4.f * zz - yy) +
                                   SH_C3[5] * s.c[14] * 2.f * xz + SH_C3[6] * s.c[15] * 3.f * (xx - yy));
                dRGBdy = dRGBdy + (SH_C3[0] * s.c[9] * 3.f * (xx - yy) + SH_C3[1] * s.c[10] * xz +
                                   SH_C3[2] * s.c[11] * (-3.f * yy + 4.f * zz - xx) + SH_C3[3] * s.c[12] * -3.f * 2.f * yz +
                                   SH_C3[4] * s.c[13] * -2.f * xy + SH_C3[5] * s.c[14] * -2.f * yz +
                                   SH_C3[6] * s.c[15] * -3.f * 2.f * xy);
                dRGBdz = dRGBdz + (SH_C3[1] * s.c[10] * xy + SH_C3[2] * s.c[11] * 4.f * 2.f * yz +
                                   SH_C3[3] * s.c[12] * 3.f * (2.f * zz - xx - yy) + SH_C3[4] * s.c[13] * 4.f * 2.f * xz +
                                   SH_C3[5] * s.c[14] * (xx - yy));
            }
        }
    }
}

// ---- per-view pieces of preprocessCUDA, shared by the one-view kernel and the batched one ----
struct FwdGeom {
    int radius;
    ushort4 rect;
    bool violation, need_sh;
};
// Frustum test, covariance projection, radius, tile rectangle and the first two pieces of the render record of Gaussian
// idx for the camera in `a` (forward.cu:155-256 up to the colour).  C3_GIVEN: the 3D covariance -- it depends on scale
// and rotation only -- was formed by the caller (the batched kernel evaluates it once for all views of the batch; same
// expression, same bits) and is only stored here.
template <bool C3_GIVEN>
__device__ __forceinline__ FwdGeom fwd_view_geometry(const PreprocessFwdArgs& a, int idx, float3 p_orig, const float (&c3_in)[6]) {
    int radius = 0;
    ushort4 rect = make_ushort4(0, 0, 0, 0);
    bool violation = false, need_sh = false;
    if (a.gau_uncertainty) a.gau_uncertainty[idx] = 0.0f;
    if (a.gau_related_pixels) a.gau_related_pixels[idx] = 0;
    // in_frustum (cuda_rasterizer/auxiliary.h:139-164)
    const float4 p_hom = xform4x4(p_orig, a.proj);
    const float p_w = 1.0f / (p_hom.w + 0.0000001f);
    const float3 p_proj = make_float3(p_hom.x * p_w, p_hom.y * p_w, p_hom.z * p_w);
    const float3 p_view = xform4x3(p_orig, a.view);
    bool live = !(p_view.z <= DGR_NEAR);
    violation = !live && a.prefiltered;  // `prefiltered` promised that nothing is culled (auxiliary.h:154-160)

    if (live) {
        float c3[6];
        if (C3_GIVEN) {
#pragma unroll
            for (int i = 0; i < 6; i++) c3[i] = c3_in[i];
        } else if (a.cov3D_precomp) {
#pragma unroll
            for (int i = 0; i < 6; i++) c3[i] = a.cov3D_precomp[6 * (size_t)idx + i];
        } else {
            // (not kept for the backward: it reads scale and rotation anyway and re-forms the covariance with the same
            //  expression, hence the same bits -- 24 bytes less written here and read there per Gaussian)
            compute_cov3d(a.scales, a.rotations, a.scale_modifier, idx, c3);
        }
        Cov2D c;
        cov2d_common(p_orig, a.focal_x, a.focal_y, a.tan_fovx, a.tan_fovy, c3, a.view, c);
        const float cx = c.cov.m[0][0] + 0.3f, cy = c.cov.m[0][1], cz = c.cov.m[1][1] + 0.3f;
        const float det = (cx * cz - cy * cy);
        if (det != 0.0f) {
            const float det_inv = 1.f / det;
            const float3 conic = make_float3(cz * det_inv, -cy * det_inv, cx * det_inv);
            const float mid = 0.5f * (cx + cz);
            const float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
            const float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
            const float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
            const float pix = ndc2pix(p_proj.x, a.W), piy = ndc2pix(p_proj.y, a.H);
            int x0, y0, x1, y1;
            get_rect(pix, piy, (int)my_radius, a.grid_x, a.grid_y, x0, y0, x1, y1);
            if ((x1 - x0) * (y1 - y0) != 0) {
                float3 rgb;
                if (a.colors_precomp) {
                    rgb = make_float3(a.colors_precomp[3 * (size_t)idx], a.colors_precomp[3 * (size_t)idx + 1],
                                      a.colors_precomp[3 * (size_t)idx + 2]);
                } else {
                    need_sh = true;  // evaluated below, after the geometry: a whole block then fetches its SH rows together
                    rgb = make_float3(0.f, 0.f, 0.f);
                }
                radius = (int)my_radius;
                int tx0 = x0, ty0 = y0, tx1 = x1, ty1 = y1;  // the rectangle that is binned (radii stay the reference's)
                if (a.tight_cull) {
                    // Optional (NOT the reference's integer path): shrink the tile rectangle to the box of
                    // the region where alpha can reach 15/255, q(d) <= tau = 2 ln(255 o / 15): half extents sqrt(tau cov_xx),
                    // sqrt(tau cov_yy) <= 2.38 sigma instead of the 3 sigma_max circle, with the same safety margin the
                    // blend kernels' own culling uses.  Every dropped (tile, Gaussian) instance is one no pixel would blend,
                    // so images and gradients are unchanged; num_rendered, the tile lists and n_contrib shrink.
                    const float o = a.opacities[idx];
                    const float tau = 2.0f * __logf(o * (255.0f / 15.0f));
                    if (!(tau > 0.0f)) {
                        tx1 = tx0;  // can never contribute
                    } else {
                        const float hx = sqrtf(tau * cx) * 1.001f + 0.05f, hy = sqrtf(tau * cz) * 1.001f + 0.05f;
                        // tile t holds pixels 16 t .. 16 t + 15 (pixel centres at integer coordinates)
                        tx0 = max(x0, (int)ceilf((pix - hx - 15.0f) / 16.0f));
                        ty0 = max(y0, (int)ceilf((piy - hy - 15.0f) / 16.0f));
                        tx1 = min(x1, (int)floorf((pix + hx) / 16.0f) + 1);
                        ty1 = min(y1, (int)floorf((piy + hy) / 16.0f) + 1);
                        if (tx1 < tx0) tx1 = tx0;
                        if (ty1 < ty0) ty1 = ty0;
                    }
                }
                rect = make_ushort4((unsigned short)tx0, (unsigned short)ty0, (unsigned short)tx1, (unsigned short)ty1);
                a.geom.depths[idx] = p_view.z;
                float4* rec = a.geom.rec + DGR_REC_STRIDE * (size_t)idx;
                rec[0] = make_float4(pix, piy, p_view.z, a.opacities[idx]);
                rec[1] = make_float4(conic.x, conic.y, conic.z, 0.0f);
                if (!need_sh) rec[2] = make_float4(rgb.x, rgb.y, rgb.z, 0.0f);
            }
        }
    }
    a.geom.radii[idx] = radius;
    if (a.radii_out) a.radii_out[idx] = radius;
    a.geom.rect[idx] = rect;
    return FwdGeom{radius, rect, violation, need_sh};
}

// computeColorFromSH (forward.cu:20-71) for the camera in `a`, plus what the backward keeps of it; writes the third piece
// of the render record
__device__ __forceinline__ void fwd_view_colour(const PreprocessFwdArgs& a, int idx, float3 p_orig, const SHCoeffs& s) {
    float3 rgb;
    const float3 cam = make_float3(a.campos[0], a.campos[1], a.campos[2]);
    float3 dir = p_orig - cam;
    const float len = sqrtf(dot3(dir, dir));
    dir = make_float3(dir.x / len, dir.y / len, dir.z / len);
    float3 res = SH_C0 * s.c[0];
    if (a.D > 0) {
        const float x = dir.x, y = dir.y, z = dir.z;
        res = res - SH_C1 * y * s.c[1] + SH_C1 * z * s.c[2] - SH_C1 * x * s.c[3];
        if (a.D > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            res = res + SH_C2[0] * xy * s.c[4] + SH_C2[1] * yz * s.c[5] +
                  SH_C2[2] * (2.0f * zz - xx - yy) * s.c[6] + SH_C2[3] * xz * s.c[7] +
                  SH_C2[4] * (xx - yy) * s.c[8];
            if (a.D > 2) {
                res = res + SH_C3[0] * y * (3.0f * xx - yy) * s.c[9] + SH_C3[1] * xy * z * s.c[10] +
                      SH_C3[2] * y * (4.0f * zz - xx - yy) * s.c[11] +
                      SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * s.c[12] +
                      SH_C3[4] * x * (4.0f * zz - xx - yy) * s.c[13] +
                      SH_C3[5] * z * (xx - yy) * s.c[14] + SH_C3[6] * x * (xx - 3.0f * yy) * s.c[15];
            }
        }
    }
    res.x += 0.5f; res.y += 0.5f; res.z += 0.5f;
    {
        // d(colour)/d(direction) (L/cuda_rasterizer/backward.cu:50-133), kept for the backward: it is all the
        // backward needs of the SH coefficients beyond the basis values, which depend on the direction alone
        float3 dRGBdx = make_float3(0, 0, 0), dRGBdy = make_float3(0, 0, 0), dRGBdz = make_float3(0, 0, 0);
        sh_direction_derivatives(s, a.D, dir, dRGBdx, dRGBdy, dRGBdz);
        float4* shd = a.geom.shd + (size_t)idx;  // three planes of P float4: consecutive lanes store consecutive 16-byte pieces
        shd[0] = make_float4(dRGBdx.x, dRGBdx.y, dRGBdx.z, 0.0f);
        shd[(size_t)a.P] = make_float4(dRGBdy.x, dRGBdy.y, dRGBdy.z, 0.0f);
        shd[2 * (size_t)a.P] = make_float4(dRGBdz.x, dRGBdz.y, dRGBdz.z, 0.0f);
    }
    a.geom.clamped[idx] = (uint8_t)((res.x < 0 ? 1 : 0) | (res.y < 0 ? 2 : 0) | (res.z < 0 ? 4 : 0));
    rgb = make_float3(fmaxf(res.x, 0.0f), fmaxf(res.y, 0.0f), fmaxf(res.z, 0.0f));
    a.geom.rec[DGR_REC_STRIDE * (size_t)idx + 2] = make_float4(rgb.x, rgb.y, rgb.z, 0.0f);
}

// ------------------------------------------------------------------------------------------------
#ifndef DGR_PPF_WAVES
#define DGR_PPF_WAVES 5
#endif
__global__ void __launch_bounds__(256, DGR_PPF_WAVES) preprocess_fwd_kernel(PreprocessFwdArgs a) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    FwdGeom g{0, make_ushort4(0, 0, 0, 0), false, false};
    float3 p_orig = make_float3(0.f, 0.f, 0.f);
    // LDS: the SH transposition buffer and, afterwards, the rank stage of the fused count share one pool
    constexpr int POOL_WORDS = (4 * SHT_ROWS * SHT_LD > COUNT_STAGE) ? 4 * SHT_ROWS * SHT_LD : COUNT_STAGE;
    __shared__ float pool[POOL_WORDS];
    // Zeroing that would otherwise be stream memsets (one launch each): the tile counters count_rank increments
    // (callback path only -- the fused count needs them cleared before this kernel starts) and the two per-Gaussian
    // median statistics the forward blend accumulates into.
    for (int i = idx; i < a.n_zero_words; i += gridDim.x * 256) a.zero_words[i] = 0u;
    if (idx < a.P) {
        p_orig = make_float3(a.means3D[3 * idx], a.means3D[3 * idx + 1], a.means3D[3 * idx + 2]);
        const float no_c3[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        g = fwd_view_geometry<false>(a, idx, p_orig, no_c3);
    }
    const bool need_sh = g.need_sh, violation = g.violation;
    const ushort4 rect = g.rect;

    // computeColorFromSH (forward.cu:20-71) for the Gaussians that survived
    if (a.shs && !a.colors_precomp) {  // uniform
        float* sht = pool;
        // whole block in range, 16 coefficients, 16-byte aligned rows: SH rows move through LDS (block-uniform)
        const bool blk_fast = a.sh_vec_ok && a.M == 16 && (size_t)blockIdx.x * 256 + 256 <= (size_t)a.P &&
                              __syncthreads_or(need_sh);
        float shf[48];
        if (blk_fast) sh_rows_to_lanes(a.shs, (size_t)blockIdx.x * 256, sht, shf);
        if (need_sh) {
            SHCoeffs s;
            if (blk_fast) {
#pragma unroll
                for (int k = 0; k < 16; k++) s.c[k] = make_float3(shf[3 * k], shf[3 * k + 1], shf[3 * k + 2]);
            } else {
                load_sh(a.shs, idx, a.D, a.M, a.sh_vec_ok, s);
            }
            fwd_view_colour(a, idx, p_orig, s);
        }
    }

    __shared__ uint32_t wsum[4];
    if (a.fused_count) {
        // duplicateWithKeys' first half, here: this block's instances get a contiguous run of the rank array handed out
        // by a global cursor (any unique placement will do -- the ranks are read back through goff), then every
        // instance takes its tile-counter atomic (count_rank.h).
        __shared__ uint32_t s_base;
        const bool any_violation = __syncthreads_or(violation);  // (also orders the SH phase's LDS reads before the stage)
        const uint32_t n = (uint32_t)(rect.z - rect.x) * (uint32_t)(rect.w - rect.y);
        uint32_t block_total;
        const uint32_t loc = block_exclusive_scan(n, wsum, threadIdx.x, &block_total);
        if (threadIdx.x == 0) {
            s_base = block_total ? atomicAdd(a.cursor, block_total) : 0u;
            if (any_violation) atomicOr(a.cursor + 1, 1u);
        }
        __syncthreads();
        const uint32_t base = s_base;
        if (idx < a.P) a.geom.goff[idx] = base + loc;
        count_and_rank(rect, base + loc, base, block_total, a.tile_count, a.ranks, a.grid_x, a.capacity,
                       reinterpret_cast<uint32_t*>(pool), threadIdx.x);
        return;
    }
    // callback path: instances (tiles_touched) of this block, for count_rank's offsets (the reference scans
    // tiles_touched over P, L/cuda_rasterizer/rasterizer_impl.cu:283)
    uint32_t n = (uint32_t)(rect.z - rect.x) * (uint32_t)(rect.w - rect.y);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) n += __shfl_xor(n, off, 64);
    // bit 31 of the block total carries "some Gaussian of this block violated `prefiltered`" (scan_blocks moves it
    // into status[2]); block totals stay far below 2^31
    if (__builtin_amdgcn_ballot_w64(violation) != 0ull) n |= 0x80000000u;
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = n;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t flag = (wsum[0] | wsum[1] | wsum[2] | wsum[3]) & 0x80000000u;
        a.geom.block_tiles[blockIdx.x] = ((wsum[0] & 0x7fffffffu) + (wsum[1] & 0x7fffffffu) + (wsum[2] & 0x7fffffffu) +
                                          (wsum[3] & 0x7fffffffu)) | flag;
    }
}

// ------------------------------------------------------------------------------------------------
// Batched forward preprocess: the V views of a batch in ONE launch.  A Gaussian's position, opacity and
// 3D covariance are fetched / formed once, its 192-byte SH row is fetched once -- when at least one view sees it -- and
// evaluated for every camera that does; per view the lane runs exactly the code of the one-view kernel
// (fwd_view_geometry / fwd_view_colour), so every view's state buffers are bit-identical to a one-view call.  Only the
// LDS-count form of the epilogue exists here (per-block instance totals in each view's geom.block_tiles).
__device__ __forceinline__ PreprocessFwdArgs batch_view_args(const PreprocessFwdBatchArgs& b, int v) {
    PreprocessFwdArgs a = b.base;
    const FwdViewPart& p = b.v[v];
    a.view = p.view; a.proj = p.proj; a.campos = p.campos; a.geom = p.geom; a.radii_out = p.radii_out;
    a.gau_uncertainty = p.gau_uncertainty; a.gau_related_pixels = p.gau_related_pixels;
    return a;
}
__global__ void __launch_bounds__(256) preprocess_fwd_batch_kernel(PreprocessFwdBatchArgs b) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int P = b.base.P, V = b.V;
    __shared__ float pool[4 * SHT_ROWS * SHT_LD];
    __shared__ uint32_t wsum[DGR_MAX_BATCH_VIEWS][4];
    float3 p_orig = make_float3(0.f, 0.f, 0.f);
    float c3[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (idx < P) {
        p_orig = make_float3(b.base.means3D[3 * idx], b.base.means3D[3 * idx + 1], b.base.means3D[3 * idx + 2]);
        if (b.base.cov3D_precomp) {
#pragma unroll
            for (int i = 0; i < 6; i++) c3[i] = b.base.cov3D_precomp[6 * (size_t)idx + i];
        } else {
            compute_cov3d(b.base.scales, b.base.rotations, b.base.scale_modifier, idx, c3);
        }
    }
    uint32_t need_mask = 0u;
#pragma unroll 1
    for (int v = 0; v < V; v++) {
        const PreprocessFwdArgs a = batch_view_args(b, v);
        FwdGeom g{0, make_ushort4(0, 0, 0, 0), false, false};
        if (idx < P) g = fwd_view_geometry<true>(a, idx, p_orig, c3);
        if (g.need_sh) need_mask |= 1u << v;
        // instances (tiles_touched) of this block in view v, as the one-view kernel leaves them for count_lds / scan_table
        uint32_t n = (uint32_t)(g.rect.z - g.rect.x) * (uint32_t)(g.rect.w - g.rect.y);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) n += __shfl_xor(n, off, 64);
        if (__builtin_amdgcn_ballot_w64(g.violation) != 0ull) n |= 0x80000000u;
        if ((threadIdx.x & 63) == 0) wsum[v][threadIdx.x >> 6] = n;
    }
    __syncthreads();
    if ((int)threadIdx.x < V) {
        const uint32_t* w = wsum[threadIdx.x];
        const uint32_t flag = (w[0] | w[1] | w[2] | w[3]) & 0x80000000u;
        b.v[threadIdx.x].geom.block_tiles[blockIdx.x] =
            ((w[0] & 0x7fffffffu) + (w[1] & 0x7fffffffu) + (w[2] & 0x7fffffffu) + (w[3] & 0x7fffffffu)) | flag;
    }
    if (b.base.shs && !b.base.colors_precomp) {  // uniform
        const bool blk_fast = b.base.sh_vec_ok && b.base.M == 16 && (size_t)blockIdx.x * 256 + 256 <= (size_t)P &&
                              __syncthreads_or(need_mask != 0u);
        float shf[48];
        if (blk_fast) sh_rows_to_lanes(b.base.shs, (size_t)blockIdx.x * 256, pool, shf);
        if (need_mask != 0u) {
            SHCoeffs s;
            if (blk_fast) {
#pragma unroll
                for (int k = 0; k < 16; k++) s.c[k] = make_float3(shf[3 * k], shf[3 * k + 1], shf[3 * k + 2]);
            } else {
                load_sh(b.base.shs, idx, b.base.D, b.base.M, b.base.sh_vec_ok, s);
            }
#pragma unroll 1
            for (int v = 0; v < V; v++)
                if ((need_mask >> v) & 1u) fwd_view_colour(batch_view_args(b, v), idx, p_orig, s);
        }
    }
}

// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) mark_visible_kernel(int P, const float* __restrict__ means,
                                                           const float* __restrict__ view, uint8_t* present) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= P) return;
    const float3 p = make_float3(means[3 * idx], means[3 * idx + 1], means[3 * idx + 2]);
    present[idx] = !(xform4x3(p, view).z <= DGR_NEAR);
}

// ------------------------------------------------------------------------------------------------
// ---- per-view terms of the fused per-Gaussian backward, shared by the one-view kernel and the batched one ----
// Everything that depends on the camera: the median-depth term, computeCov2DCUDA, preprocessCUDA, the SH backward up to
// the scalars coef[k] and the masked colour gradient (dL_dsh[k] = coef[k] * dRGB), the pose-gradient terms.  `acc` = the
// blend backward's sums for this Gaussian in this view (zeros when it was not visible).
__device__ __forceinline__ void bwd_view_terms(const PreprocessBwdArgs& a, const bool full, float3 m, const float (&c3)[6],
                                               const float (&acc)[16], bool vis, uint8_t cl_in, float4 shd0, float4 shd1,
                                               float4 shd2, float3& dmean_out, float (&dcov)[6], float (&coef)[16],
                                               float3& dRGB, float (&pose)[12]) {
    // light: the blend kernel's median-depth term; full: computeCov2DCUDA ASSIGNS (F/cuda_rasterizer/backward.cu:383)
    float3 dmean = make_float3(0.f, 0.f, 0.f);
    if (!full) {
        // light: the blend kernel's median-depth term (L/cuda_rasterizer/backward.cu:654-664), whose pixel sum of
        // dL/dmedian arrives in acc[10]; the per-Gaussian factors are applied here
        const float* v = a.view;
        const float mul3 = v[2] * m.x + v[6] * m.y + v[10] * m.z + v[14];
        dmean = make_float3((v[2] - v[3] * mul3) * acc[10], (v[6] - v[7] * mul3) * acc[10], (v[10] - v[11] * mul3) * acc[10]);
    }
    float3 s_cam = make_float3(0.f, 0.f, 0.f);  // full: sum_ch dL_dcolor[ch] * d(rgb[ch])/d(campos.{x,y,z})
    const bool do_map = vis && !a.map_off;
#pragma unroll
    for (int i = 0; i < 6; i++) dcov[i] = 0.0f;
    if (do_map) {
        // ---------------- computeCov2DCUDA (L/cuda_rasterizer/backward.cu:144-276)
        const float3 dconic = make_float3(acc[6], acc[7], acc[8]);
        Cov2D c;
        cov2d_common(m, a.focal_x, a.focal_y, a.tan_fovx, a.tan_fovy, c3, a.view, c);
        const float limx = 1.3f * a.tan_fovx, limy = 1.3f * a.tan_fovy;
        const float x_grad_mul = (c.txtz < -limx || c.txtz > limx) ? 0.f : 1.f;
        const float y_grad_mul = (c.tytz < -limy || c.tytz > limy) ? 0.f : 1.f;
        const float h_x = a.focal_x, h_y = a.focal_y;
        const M3& T = c.T; const M3& Vrk = c.Vrk; const M3& W = c.W; const float3 t = c.t;
        const float ca = c.cov.m[0][0] + 0.3f, cb = c.cov.m[0][1], cc = c.cov.m[1][1] + 0.3f;
        const float denom = ca * cc - cb * cb;
        float dL_da = 0, dL_db = 0, dL_dc = 0;
        const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
        if (denom2inv != 0) {
            dL_da = denom2inv * (-cc * cc * dconic.x + 2 * cb * cc * dconic.y + (denom - ca * cc) * dconic.z);
            dL_dc = denom2inv * (-ca * ca * dconic.z + 2 * ca * cb * dconic.y + (denom - ca * cc) * dconic.x);
            dL_db = denom2inv * 2 * (cb * cc * dconic.x - (denom + 2 * cb * cb) * dconic.y + ca * cb * dconic.z);
            dcov[0] = (T.m[0][0] * T.m[0][0] * dL_da + T.m[0][0] * T.m[1][0] * dL_db + T.m[1][0] * T.m[1][0] * dL_dc);
            dcov[3] = (T.m[0][1] * T.m[0][1] * dL_da + T.m[0][1] * T.m[1][1] * dL_db + T.m[1][1] * T.m[1][1] * dL_dc);
            dcov[5] = (T.m[0][2] * T.m[0][2] * dL_da + T.m[0][2] * T.m[1][2] * dL_db + T.m[1][2] * T.m[1][2] * dL_dc);
            dcov[1] = 2 * T.m[0][0] * T.m[0][1] * dL_da + (T.m[0][0] * T.m[1][1] + T.m[0][1] * T.m[1][0]) * dL_db + 2 * T.m[1][0] * T.m[1][1] * dL_dc;
            dcov[2] = 2 * T.m[0][0] * T.m[0][2] * dL_da + (T.m[0][0] * T.m[1][2] + T.m[0][2] * T.m[1][0]) * dL_db + 2 * T.m[1][0] * T.m[1][2] * dL_dc;
            dcov[4] = 2 * T.m[0][2] * T.m[0][1] * dL_da + (T.m[0][1] * T.m[1][2] + T.m[0][2] * T.m[1][1]) * dL_db + 2 * T.m[1][1] * T.m[1][2] * dL_dc;
        }
        const float dL_dT00 = 2 * (T.m[0][0] * Vrk.m[0][0] + T.m[0][1] * Vrk.m[0][1] + T.m[0][2] * Vrk.m[0][2]) * dL_da +
                              (T.m[1][0] * Vrk.m[0][0] + T.m[1][1] * Vrk.m[0][1] + T.m[1][2] * Vrk.m[0][2]) * dL_db;
        const float dL_dT01 = 2 * (T.m[0][0] * Vrk.m[1][0] + T.m[0][1] * Vrk.m[1][1] + T.m[0][2] * Vrk.m[1][2]) * dL_da +
                              (T.m[1][0] * Vrk.m[1][0] + T.m[1][1] * Vrk.m[1][1] + T.m[1][2] * Vrk.m[1][2]) * dL_db;
        const float dL_dT02 = 2 * (T.m[0][0] * Vrk.m[2][0] + T.m[0][1] * Vrk.m[2][1] + T.m[0][2] * Vrk.m[2][2]) * dL_da +
                              (T.m[1][0] * Vrk.m[2][0] + T.m[1][1] * Vrk.m[2][1] + T.m[1][2] * Vrk.m[2][2]) * dL_db;
        const float dL_dT10 = 2 * (T.m[1][0] * Vrk.m[0][0] + T.m[1][1] * Vrk.m[0][1] + T.m[1][2] * Vrk.m[0][2]) * dL_dc +
                              (T.m[0][0] * Vrk.m[0][0] + T.m[0][1] * Vrk.m[0][1] + T.m[0][2] * Vrk.m[0][2]) * dL_db;
        const float dL_dT11 = 2 * (T.m[1][0] * Vrk.m[1][0] + T.m[1][1] * Vrk.m[1][1] + T.m[1][2] * Vrk.m[1][2]) * dL_dc +
                              (T.m[0][0] * Vrk.m[1][0] + T.m[0][1] * Vrk.m[1][1] + T.m[0][2] * Vrk.m[1][2]) * dL_db;
        const float dL_dT12 = 2 * (T.m[1][0] * Vrk.m[2][0] + T.m[1][1] * Vrk.m[2][1] + T.m[1][2] * Vrk.m[2][2]) * dL_dc +
                              (T.m[0][0] * Vrk.m[2][0] + T.m[0][1] * Vrk.m[2][1] + T.m[0][2] * Vrk.m[2][2]) * dL_db;
        const float dL_dJ00 = W.m[0][0] * dL_dT00 + W.m[0][1] * dL_dT01 + W.m[0][2] * dL_dT02;
        const float dL_dJ02 = W.m[2][0] * dL_dT00 + W.m[2][1] * dL_dT01 + W.m[2][2] * dL_dT02;
        const float dL_dJ11 = W.m[1][0] * dL_dT10 + W.m[1][1] * dL_dT11 + W.m[1][2] * dL_dT12;
        const float dL_dJ12 = W.m[2][0] * dL_dT10 + W.m[2][1] * dL_dT11 + W.m[2][2] * dL_dT12;
        const float tz = 1.f / t.z;
        const float tz2 = tz * tz;
        const float tz3 = tz2 * tz;
        const float dL_dtx = x_grad_mul * -h_x * tz2 * dL_dJ02;
        const float dL_dty = y_grad_mul * -h_y * tz2 * dL_dJ12;
        const float dL_dtz = -h_x * tz2 * dL_dJ00 - h_y * tz2 * dL_dJ11 + (2 * h_x * t.x) * tz3 * dL_dJ02 + (2 * h_y * t.y) * tz3 * dL_dJ12;
        const float* v = a.view;
        dmean.x += v[0] * dL_dtx + v[1] * dL_dty + v[2] * dL_dtz;
        dmean.y += v[4] * dL_dtx + v[5] * dL_dty + v[6] * dL_dtz;
        dmean.z += v[8] * dL_dtx + v[9] * dL_dty + v[10] * dL_dtz;
        if (full) {  // depth -> mean term inside computeCov2DCUDA (F/cuda_rasterizer/backward.cu:385-386)
            const float mul3f = v[2] * m.x + v[6] * m.y + v[10] * m.z + v[14];
            dmean.x = dmean.x + acc[3] * (v[2] - v[3] * mul3f);
            dmean.y = dmean.y + acc[3] * (v[6] - v[7] * mul3f);
            dmean.z = dmean.z + acc[3] * (v[10] - v[11] * mul3f);
        }

        // ---------------- preprocessCUDA (L/cuda_rasterizer/backward.cu:348-416)
        const float* pj = a.proj;
        const float4 m_hom = xform4x4(m, pj);
        const float m_w = 1.0f / (m_hom.w + 0.0000001f);
        const float mul1 = (pj[0] * m.x + pj[4] * m.y + pj[8] * m.z + pj[12]) * m_w * m_w;
        const float mul2 = (pj[1] * m.x + pj[5] * m.y + pj[9] * m.z + pj[13]) * m_w * m_w;
        const float g2x = acc[4], g2y = acc[5];
        float3 d1;
        d1.x = (pj[0] * m_w - pj[3] * mul1) * g2x + (pj[1] * m_w - pj[3] * mul2) * g2y;
        d1.y = (pj[4] * m_w - pj[7] * mul1) * g2x + (pj[5] * m_w - pj[7] * mul2) * g2y;
        d1.z = (pj[8] * m_w - pj[11] * mul1) * g2x + (pj[9] * m_w - pj[11] * mul2) * g2y;
        dmean.x += d1.x; dmean.y += d1.y; dmean.z += d1.z;
        if (!full) {  // light: depth -> mean term inside preprocessCUDA (L/cuda_rasterizer/backward.cu:396-407)
            const float mul3 = v[2] * m.x + v[6] * m.y + v[10] * m.z + v[14];
            float3 d2;
            d2.x = (v[2] - v[3] * mul3) * acc[3];
            d2.y = (v[6] - v[7] * mul3) * acc[3];
            d2.z = (v[10] - v[11] * mul3) * acc[3];
            dmean.x += d2.x; dmean.y += d2.y; dmean.z += d2.z;
        }
    }
    // ---------------- SH backward (L/cuda_rasterizer/backward.cu:20-139): the scalars and the masked colour gradient
#pragma unroll
    for (int k = 0; k < 16; k++) coef[k] = 0.0f;
    dRGB = make_float3(0.f, 0.f, 0.f);
    if (a.dL_dsh && a.M > 0 && do_map && a.shs) {
        const float3 cam = make_float3(a.campos[0], a.campos[1], a.campos[2]);
        const float3 dir_orig = m - cam;
        const float len = sqrtf(dot3(dir_orig, dir_orig));
        const float3 dir = make_float3(dir_orig.x / len, dir_orig.y / len, dir_orig.z / len);
        const uint8_t cl = cl_in;
        dRGB = make_float3(acc[0], acc[1], acc[2]);
        dRGB.x *= (cl & 1) ? 0 : 1;
        dRGB.y *= (cl & 2) ? 0 : 1;
        dRGB.z *= (cl & 4) ? 0 : 1;
        // d(colour)/d(direction): the forward evaluated it from the SH row it had in registers (geom.shd, requested
        // with the other inputs above) -- the basis values below need the direction only, so the 192-byte rows are
        // not read again
        const float3 dRGBdx = make_float3(shd0.x, shd0.y, shd0.z), dRGBdy = make_float3(shd1.x, shd1.y, shd1.z),
                     dRGBdz = make_float3(shd2.x, shd2.y, shd2.z);
        const float x = dir.x, y = dir.y, z = dir.z;
        coef[0] = SH_C0;
        if (a.D > 0) {
            coef[1] = -SH_C1 * y;
            coef[2] = SH_C1 * z;
            coef[3] = -SH_C1 * x;
            if (a.D > 1) {
                const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                coef[4] = SH_C2[0] * xy;
                coef[5] = SH_C2[1] * yz;
                coef[6] = SH_C2[2] * (2.f * zz - xx - yy);
                coef[7] = SH_C2[3] * xz;
                coef[8] = SH_C2[4] * (xx - yy);
                if (a.D > 2) {
                    coef[9] = SH_C3[0] * y * (3.f * xx - yy);
                    coef[10] = SH_C3[1] * xy * z;
                    coef[11] = SH_C3[2] * y * (4.f * zz - xx - yy);
                    coef[12] = SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy);
                    coef[13] = SH_C3[4] * x * (4.f * zz - xx - yy);
                    coef[14] = SH_C3[5] * z * (xx - yy);
                    coef[15] = SH_C3[6] * x * (xx - 3.f * yy);
                }
            }
        }
        if (full) {
            // dgc_dCampos (F/cuda_rasterizer/backward.cu:27-43,159-166; not clamp-masked) contracted with the
            // raw colour gradient: all ComputePG's part 1 needs of this Gaussian (:990-1022, 1313-1324)
            const float len3 = len * len * len;
            const float i3 = 1.0f / len3, i1 = 1.0f / len;
            const float3 o = dir_orig;
            const float3 raw = make_float3(acc[0], acc[1], acc[2]);
            const float3 cx = dRGBdx * (o.x * o.x * i3 - i1) + dRGBdy * (o.x * o.y * i3) + dRGBdz * (o.x * o.z * i3);
            const float3 cy = dRGBdx * (o.x * o.y * i3) + dRGBdy * (o.y * o.y * i3 - i1) + dRGBdz * (o.y * o.z * i3);
            const float3 cz = dRGBdx * (o.x * o.z * i3) + dRGBdy * (o.y * o.z * i3) + dRGBdz * (o.z * o.z * i3 - i1);
            s_cam = make_float3(dot3(raw, cx), dot3(raw, cy), dot3(raw, cz));
        }
        const float3 dL_ddir = make_float3(dot3(dRGBdx, dRGB), dot3(dRGBdy, dRGB), dot3(dRGBdz, dRGB));
        // dnormvdv (cuda_rasterizer/auxiliary.h:109-119)
        {
            const float3 vv = dir_orig, dv = dL_ddir;
            const float sum2 = vv.x * vv.x + vv.y * vv.y + vv.z * vv.z;
            const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
            dmean.x += ((+sum2 - vv.x * vv.x) * dv.x - vv.y * vv.x * dv.y - vv.z * vv.x * dv.z) * invsum32;
            dmean.y += (-vv.x * vv.y * dv.x + (sum2 - vv.y * vv.y) * dv.y - vv.z * vv.y * dv.z) * invsum32;
            dmean.z += (-vv.x * vv.z * dv.x - vv.y * vv.z * dv.y + (sum2 - vv.z * vv.z) * dv.z) * invsum32;
        }
    }
    // ---------------- pose gradient: sum over Gaussians of Jacobian x (sum over pixels)
    // L/cuda_rasterizer/backward.cu:633-651 accumulates J_k(g) * {nx, ny, dL_ddepth} per pixel; J_k
    // depends on the Gaussian only, so the pixel sums are taken first (acc[4], acc[5], acc[13]).
    if (vis && !a.track_off) {
        const float4 m_hom = xform4x4(m, a.proj);
        const float m_w = 1.0f / (m_hom.w + 0.0000001f);
        const float mm[4] = {m.x, m.y, m.z, 1.0f};
        if (!full) {
            const float A = acc[4], B = acc[5], Dd = acc[13];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                pose[3 * k + 0] = (m_w * a.perspec[0] * mm[k]) * A;
                pose[3 * k + 1] = (m_w * a.perspec[5] * mm[k]) * B;
                pose[3 * k + 2] = (m_hom.x * (-m_w * m_w) * mm[k]) * A + (m_hom.y * (-m_w * m_w) * mm[k]) * B + mm[k] * Dd;
            }
        } else {
            // ComputePG (F/cuda_rasterizer/backward.cu:990-1072, 1247-1289, 1313-1324) summed per Gaussian:
            // part 1 (colour -> campos -> view) + part 2-1 (ndc -> view, colour terms) + the depth terms of the
            // pixels whose front-most valid Gaussian this is.
            const float A = acc[10], B = acc[11], Dw = acc[12], Dx = acc[13], Dy = acc[14];
            const float* v = a.view;
            const float sc[3] = {s_cam.x, s_cam.y, s_cam.z};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const float jx0 = m_w * a.perspec[0] * mm[k], jy1 = m_w * a.perspec[5] * mm[k];
                const float jx2 = m_hom.x * (-m_w * m_w) * mm[k], jy2 = m_hom.y * (-m_w * m_w) * mm[k];
#pragma unroll
                for (int j = 0; j < 3; j++) {
                    float p1;
                    if (k < 3) p1 = sc[k] * (-v[12 + j]);
                    else p1 = sc[0] * (-v[j]) + sc[1] * (-v[4 + j]) + sc[2] * (-v[8 + j]);
                    float p21, dpt;
                    if (j == 0) { p21 = jx0 * A; dpt = jx0 * Dx; }
                    else if (j == 1) { p21 = jy1 * B; dpt = jy1 * Dy; }
                    else { p21 = jx2 * A + jy2 * B; dpt = mm[k] * Dw + (jx2 * Dx + jy2 * Dy); }
                    pose[3 * k + j] = (p1 + p21) + dpt;
                }
            }
        }
    }
    dmean_out = dmean;
}

// computeCov3D backward (L/cuda_rasterizer/backward.cu:280-343): linear in dL_dcov3D
__device__ __forceinline__ void cov3d_backward_terms(float3 sc, float4 q, float mod, const float (&dcov)[6], float3& dscale, float4& drot) {
    const float r = q.x, x = q.y, y = q.z, z = q.w;
    M3 R;
    quat_to_R(q, R);
    const float3 s = make_float3(mod * sc.x, mod * sc.y, mod * sc.z);
    const M3 Mm = mul(diag3(s.x, s.y, s.z), R);
    M3 dSigma;
    dSigma.m[0][0] = dcov[0]; dSigma.m[0][1] = 0.5f * dcov[1]; dSigma.m[0][2] = 0.5f * dcov[2];
    dSigma.m[1][0] = 0.5f * dcov[1]; dSigma.m[1][1] = dcov[3]; dSigma.m[1][2] = 0.5f * dcov[4];
    dSigma.m[2][0] = 0.5f * dcov[2]; dSigma.m[2][1] = 0.5f * dcov[4]; dSigma.m[2][2] = dcov[5];
    M3 M2;
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int rr = 0; rr < 3; rr++) M2.m[c][rr] = Mm.m[c][rr] * 2.0f;
    const M3 dL_dM = mul(M2, dSigma);
    const M3 Rt = transpose(R);
    M3 dMt = transpose(dL_dM);
    dscale.x = dot3(make_float3(Rt.m[0][0], Rt.m[0][1], Rt.m[0][2]), make_float3(dMt.m[0][0], dMt.m[0][1], dMt.m[0][2]));
    dscale.y = dot3(make_float3(Rt.m[1][0], Rt.m[1][1], Rt.m[1][2]), make_float3(dMt.m[1][0], dMt.m[1][1], dMt.m[1][2]));
    dscale.z = dot3(make_float3(Rt.m[2][0], Rt.m[2][1], Rt.m[2][2]), make_float3(dMt.m[2][0], dMt.m[2][1], dMt.m[2][2]));
#pragma unroll
    for (int k = 0; k < 3; k++) { dMt.m[0][k] *= s.x; dMt.m[1][k] *= s.y; dMt.m[2][k] *= s.z; }
    drot.x = 2 * z * (dMt.m[0][1] - dMt.m[1][0]) + 2 * y * (dMt.m[2][0] - dMt.m[0][2]) + 2 * x * (dMt.m[1][2] - dMt.m[2][1]);
    drot.y = 2 * y * (dMt.m[1][0] + dMt.m[0][1]) + 2 * z * (dMt.m[2][0] + dMt.m[0][2]) + 2 * r * (dMt.m[1][2] - dMt.m[2][1]) - 4 * x * (dMt.m[2][2] + dMt.m[1][1]);
    drot.z = 2 * x * (dMt.m[1][0] + dMt.m[0][1]) + 2 * r * (dMt.m[2][0] - dMt.m[0][2]) + 2 * z * (dMt.m[1][2] + dMt.m[2][1]) - 4 * y * (dMt.m[2][2] + dMt.m[0][0]);
    drot.w = 2 * r * (dMt.m[0][1] - dMt.m[1][0]) + 2 * x * (dMt.m[2][0] + dMt.m[0][2]) + 2 * y * (dMt.m[1][2] + dMt.m[2][1]) - 4 * z * (dMt.m[1][1] + dMt.m[0][0]);
}

// Block reduction of the 12 pose terms and their delivery.  The sum over 4e5 Gaussians cancels to ~1e-3 of its terms, so
// everything beyond a 16-lane row is accumulated in double.
// The block's partial goes into one of 64 bucket rows with double atomics performed at L2 (agent scope: no
// cache to keep coherent), then the block takes a ticket; the block that draws the last ticket finds every
// partial delivered and finishes the sum -- no separate reduction kernel, no fence that writes back an L2.
template <int CTRL>
__device__ __forceinline__ float dpp_row_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
// Step 1 (every wave): the 16 lanes of a DPP row are summed in float on the vector pipe (quad_perm ^1, ^2, row_half_mirror,
// row_mirror: four adds per value, every lane of the row ends up with the row's sum) and the row sums go to LDS as doubles.
// Sixteen float terms add nothing to the rounding the float terms already carry (each is a float product), and the 144
// ds_bpermute + 72 double adds per wave of the all-double 64-lane butterfly this replaces were 11 of the one-view kernel's
// 52 us (measured with the reduction compiled out).
__device__ __forceinline__ void pose_rows_to_lds(const float (&pose)[12], double (*red)[12]) {
    const int row = threadIdx.x >> 4;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        float v = pose[i];
        v += dpp_row_mov<0xB1>(v);   // quad_perm [1,0,3,2]
        v += dpp_row_mov<0x4E>(v);   // quad_perm [2,3,0,1]
        v += dpp_row_mov<0x141>(v);  // row_half_mirror
        v += dpp_row_mov<0x140>(v);  // row_mirror
        if ((threadIdx.x & 15) == 0) red[row][i] = (double)v;
    }
}
// Step 2 (wave 0, after a workgroup barrier): the 16 row sums of the block in double, added to one of 64 bucket rows with
// double atomics performed at L2 (agent scope: no cache to keep coherent).
__device__ __forceinline__ void pose_add_partial(double (*red)[12], double* pose_part) {
    if (threadIdx.x < 12) {
        double part = 0.0;
#pragma unroll
        for (int r = 0; r < 16; r++) part += red[r][threadIdx.x];
        double* slot = pose_part + (size_t)(blockIdx.x % DGR_POSE_BUCKETS) * 12 + threadIdx.x;
        __hip_atomic_fetch_add(slot, part, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// Step 3 (wave 0, once its adds are acknowledged and it has drawn ticket `t`): the block that draws the last ticket finds
// every partial delivered and finishes the sum -- no separate reduction kernel, no fence that writes back an L2.
// `clear` (resident scratch, dgr_backward_scratch_clean_arm): the finisher leaves buckets and ticket as it found them at the start
// of the call -- zero -- for the next backward that uses this scratch.
__device__ __forceinline__ void pose_finish_if_last(uint32_t t, double* pose_part, float* dL_dview, uint32_t* ticket = nullptr,
                                                    bool clear = false) {
    if (t != gridDim.x - 1) return;
    if (threadIdx.x < 16) {
        float out = 0.0f;
        if (threadIdx.x < 12) {
            double tot = 0.0;
            for (int g = 0; g < DGR_POSE_BUCKETS; g++)  // (agent-scope loads: served by L2, where the adds were performed)
                tot += __hip_atomic_load(pose_part + (size_t)g * 12 + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            out = (float)tot;
            // (the clearing stores in a loop of their own, behind ALL the loads: a store right behind each load of the same
            //  address made the 64 round trips of this one wave dependent -- +25 us at the kernel's tail, whatever its size)
            if (clear)
                for (int g = 0; g < DGR_POSE_BUCKETS; g++)
                    __hip_atomic_store(pose_part + (size_t)g * 12 + threadIdx.x, 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (clear && threadIdx.x == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // slot order v0,v1,v2,v4,v5,v6,v8,v9,v10,v12,v13,v14 (L/cuda_rasterizer/backward.cu:723)
        if (threadIdx.x < 12) dL_dview[(threadIdx.x / 3) * 4 + threadIdx.x % 3] = out;
        if (threadIdx.x < 4) dL_dview[threadIdx.x * 4 + 3] = 0.0f;
    }
}
// One view: the delivery is wave 0's alone (no workgroup barrier after the first): two L2 round trips -- the bucket adds,
// then the ticket -- during which the other three waves would only hold their registers.
__device__ __forceinline__ void pose_block_reduce(const float (&pose)[12], double* pose_part, uint32_t* ticket, float* dL_dview,
                                                  double (*red)[12], bool clear = false) {
    pose_rows_to_lds(pose, red);
    __syncthreads();
    if (threadIdx.x >= 64) return;
    pose_add_partial(red, pose_part);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the adds are acknowledged before the ticket is taken
    uint32_t t = 0u;
    if (threadIdx.x == 0) t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    pose_finish_if_last((uint32_t)__builtin_amdgcn_readfirstlane((int)t), pose_part, dL_dview, ticket, clear);
}

// the block that drew a view's last ticket adds the per-block partials in a fixed order (wave 0 only)
__device__ __forceinline__ void pose_finish_det(const double* det_pose, uint32_t* ticket, float* dL_dview, double (*lane_sum)[12], bool clear) {
    {   // (a row's twelve loads in flight together, two rows per trip: one memory round trip per 128 rows, not per value)
        double acc[12];
#pragma unroll
        for (int c = 0; c < 12; c++) acc[c] = 0.0;
        for (uint32_t b = threadIdx.x; b < gridDim.x; b += 128) {
            double v[2][12];
            const bool two = b + 64 < gridDim.x;
#pragma unroll
            for (int c = 0; c < 12; c++) {
                v[0][c] = __hip_atomic_load(det_pose + (size_t)b * 12 + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                v[1][c] = two ? __hip_atomic_load(det_pose + (size_t)(b + 64) * 12 + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
            }
#pragma unroll
            for (int c = 0; c < 12; c++) acc[c] = (acc[c] + v[0][c]) + v[1][c];  // rows l, l + 64, l + 128, ... ascending
        }
#pragma unroll
        for (int c = 0; c < 12; c++) lane_sum[threadIdx.x][c] = acc[c];
    }
    // (one wave: LDS writes and reads of a wave are in program order)
    if (threadIdx.x < 16) {
        float out = 0.0f;
        if (threadIdx.x < 12) {
            double tot = 0.0;
            for (int l = 0; l < 64; l++) tot += lane_sum[l][threadIdx.x];
            out = (float)tot;
        }
        if (clear && threadIdx.x == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (threadIdx.x < 12) dL_dview[(threadIdx.x / 3) * 4 + threadIdx.x % 3] = out;
        if (threadIdx.x < 4) dL_dview[threadIdx.x * 4 + 3] = 0.0f;
    }
}

// Deterministic form (dgr_set_option("deterministic_grads", 1)): double atomics on the 64 bucket rows arrive in any order, and a
// double sum depends on its order in the last bit.  Here every block STORES its partial to its own row of `det_pose` and the block
// that draws the last ticket adds the rows in a fixed order: lane l of wave 0 the rows l, l + 64, ... ascending, then lanes 0..11
// the 64 lane sums ascending.
__device__ __forceinline__ void pose_block_reduce_det(const float (&pose)[12], double* det_pose, uint32_t* ticket, float* dL_dview,
                                                      double (*red)[12], bool clear) {
    __shared__ double lane_sum[64][12];
    pose_rows_to_lds(pose, red);
    __syncthreads();
    if (threadIdx.x >= 64) return;
    if (threadIdx.x < 12) {
        double part = 0.0;
#pragma unroll
        for (int r = 0; r < 16; r++) part += red[r][threadIdx.x];
        __hip_atomic_store(det_pose + (size_t)blockIdx.x * 12 + threadIdx.x, part, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the stores are acknowledged before the ticket is taken
    uint32_t t = 0u;
    if (threadIdx.x == 0) t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((uint32_t)__builtin_amdgcn_readfirstlane((int)t) != gridDim.x - 1) return;
    pose_finish_det(det_pose, ticket, dL_dview, lane_sum, clear);
}

// Fused per-Gaussian backward.  Order of the dL_dmean3D accumulation follows the reference's kernel
// order: blend-kernel median term, computeCov2DCUDA, preprocessCUDA (2D mean, depth, SH).
// (forcing more than 4 waves/SIMD spills: 5 -> 128 us, 6 -> 163 us against 87 us)
#ifndef DGR_PPB_WAVES
#define DGR_PPB_WAVES 4
#endif
__global__ void __launch_bounds__(256, DGR_PPB_WAVES) preprocess_bwd_kernel(PreprocessBwdArgs a) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    float pose[12];
#pragma unroll
    for (int i = 0; i < 12; i++) pose[i] = 0.0f;

    if (idx < a.P) {
        // Every per-Gaussian input is requested up front, unconditionally (a culled Gaussian wastes ~130 bytes): behind
        // `if (vis)` / `if (do_map)` the loads formed a chain of three dependent round trips per wave, and this kernel
        // spends 60 % of its wave-cycles waiting for memory.
        float4* ap = reinterpret_cast<float4*>(a.acc + (size_t)idx * DGR_ACC_STRIDE);
        const float4 a0 = ap[0], a1 = ap[1], a2 = ap[2], a3 = ap[3];  // (nontemporal loads here: 45 -> 49 us -- the rows sit in L2, where the blend's atomics left them)
        const int rad = a.radii[idx];

        const float3 m = make_float3(a.means3D[3 * idx], a.means3D[3 * idx + 1], a.means3D[3 * idx + 2]);
        // A tracking step (map_off: the pose gradient only) needs the three sums, the mean and the radius: the covariance, the
        // scale / rotation, the clamp bits and the SH direction derivatives -- 77 of the 157 bytes a Gaussian costs this kernel --
        // feed only the per-Gaussian gradients.  The test is kernel-uniform and known at launch: no load waits for another.
        const bool need_map = !a.map_off;
        float c3[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (need_map) {
            if (a.cov3D_precomp) {
                const float* c3p = a.cov3D_precomp + 6 * (size_t)idx;
#pragma unroll
                for (int i = 0; i < 6; i++) c3[i] = c3p[i];
            } else {  // re-formed from scale and rotation: the forward's expression, the forward's bits
                compute_cov3d(a.scales, a.rotations, a.scale_modifier, idx, c3);
            }
        }
        float3 sc_in = make_float3(0.f, 0.f, 0.f);
        float4 q_in = make_float4(0.f, 0.f, 0.f, 0.f);
        if (need_map && a.scales) {
            sc_in = make_float3(a.scales[3 * idx], a.scales[3 * idx + 1], a.scales[3 * idx + 2]);
            q_in = make_float4(a.rotations[4 * idx], a.rotations[4 * idx + 1], a.rotations[4 * idx + 2], a.rotations[4 * idx + 3]);
        }
        uint8_t cl_in = 0;
        float4 shd0 = make_float4(0.f, 0.f, 0.f, 0.f), shd1 = shd0, shd2 = shd0;
        if (need_map) {
            cl_in = a.geom.clamped[idx];
            shd0 = a.geom.shd[idx]; shd1 = a.geom.shd[(size_t)a.P + idx]; shd2 = a.geom.shd[2 * (size_t)a.P + idx];
        }
        const bool vis = rad > 0;
        float acc[16];
        if (vis) {
            acc[0] = a0.x; acc[1] = a0.y; acc[2] = a0.z; acc[3] = a0.w; acc[4] = a1.x; acc[5] = a1.y; acc[6] = a1.z; acc[7] = a1.w;
            acc[8] = a2.x; acc[9] = a2.y; acc[10] = a2.z; acc[11] = a2.w; acc[12] = a3.x; acc[13] = a3.y; acc[14] = a3.z; acc[15] = a3.w;
        } else {
#pragma unroll
            for (int i = 0; i < 16; i++) acc[i] = 0.0f;
        }

        // ---- outputs that are plain copies of the blend kernel's sums
        // (with map_off the blend kernel still sums acc[4], acc[5] for the pose gradient, but the
        //  reference leaves every per-Gaussian gradient at zero: L/cuda_rasterizer/backward.cu:666)
        // (every dense output may be NULL: a tracking step needs the pose gradient only, dgr_hip.h)
        if (a.dL_dmean2D) {
            a.dL_dmean2D[3 * (size_t)idx + 0] = a.map_off ? 0.0f : acc[4];
            a.dL_dmean2D[3 * (size_t)idx + 1] = a.map_off ? 0.0f : acc[5];
            a.dL_dmean2D[3 * (size_t)idx + 2] = 0.0f;
        }
        if (a.dL_dopacity) a.dL_dopacity[idx] = acc[9];
        if (a.dL_dcolor) {
            a.dL_dcolor[3 * (size_t)idx + 0] = acc[0];
            a.dL_dcolor[3 * (size_t)idx + 1] = acc[1];
            a.dL_dcolor[3 * (size_t)idx + 2] = acc[2];
        }
        if (a.dL_ddepth) a.dL_ddepth[idx] = acc[3];
        if (a.dL_dconic) {
            float4* o = reinterpret_cast<float4*>(a.dL_dconic) + idx;
            *o = make_float4(acc[6], acc[7], 0.0f, acc[8]);
        }

        float3 dmean;
        float dcov[6], coef[16];
        float3 dRGB;
        bwd_view_terms(a, a.full_variant != 0, m, c3, acc, vis, cl_in, shd0, shd1, shd2, dmean, dcov, coef, dRGB, pose);
        float3 dscale = make_float3(0, 0, 0);
        float4 drot = make_float4(0, 0, 0, 0);
        const bool do_map = vis && !a.map_off;
        const int ncoef_out = a.M;

        // ---------------- the dense dL_dsh row: dL_dsh[k] = coef[k] * dRGB (zeros for rows this view did not map)
        if (a.dL_dsh && ncoef_out > 0) {
            float3* out = reinterpret_cast<float3*>(a.dL_dsh) + (size_t)idx * ncoef_out;
            // whole block in range, 16 coefficients, 16-byte aligned rows: SH rows move through LDS (block-uniform)
            __shared__ float sht[4 * SHT_ROWS * SHT_LD];
            const bool blk_fast = a.sh_vec_ok && ncoef_out == 16 && a.shs != nullptr && (size_t)blockIdx.x * 256 + 256 <= (size_t)a.P;
            if (do_map && a.shs) {
                if (blk_fast) {
                    // (written below, through LDS)
                } else if (a.sh_vec_ok && ncoef_out == 16) {
                    const float ch[3] = {dRGB.x, dRGB.y, dRGB.z};
                    float4* o4 = reinterpret_cast<float4*>(out);
#pragma unroll
                    for (int i = 0; i < 12; i++)
                        o4[i] = make_float4(coef[(4 * i) / 3] * ch[(4 * i) % 3], coef[(4 * i + 1) / 3] * ch[(4 * i + 1) % 3],
                                            coef[(4 * i + 2) / 3] * ch[(4 * i + 2) % 3], coef[(4 * i + 3) / 3] * ch[(4 * i + 3) % 3]);
                } else {
#pragma unroll
                    for (int k = 0; k < 16; k++)
                        if (k < ncoef_out) out[k] = coef[k] * dRGB;
                }
            } else if (!blk_fast) {
                if (a.sh_vec_ok && ncoef_out == 16) {
                    float4* o4 = reinterpret_cast<float4*>(out);
#pragma unroll
                    for (int i = 0; i < 12; i++) o4[i] = make_float4(0, 0, 0, 0);
                } else {
                    for (int k = 0; k < ncoef_out; k++) out[k] = make_float3(0, 0, 0);
                }
            }
            if (blk_fast) lanes_to_sh_rows_scaled(coef, dRGB, a.dL_dsh, (size_t)blockIdx.x * 256, sht);
        }

        if (do_map && a.scales) cov3d_backward_terms(sc_in, q_in, a.scale_modifier, dcov, dscale, drot);

        if (a.dL_dmean3D) {
            a.dL_dmean3D[3 * (size_t)idx + 0] = dmean.x;
            a.dL_dmean3D[3 * (size_t)idx + 1] = dmean.y;
            a.dL_dmean3D[3 * (size_t)idx + 2] = dmean.z;
        }
        if (a.dL_dcov3D) {
#pragma unroll
            for (int i = 0; i < 6; i++) a.dL_dcov3D[6 * (size_t)idx + i] = dcov[i];
        }
        if (a.dL_dscale) {
            a.dL_dscale[3 * (size_t)idx + 0] = dscale.x;
            a.dL_dscale[3 * (size_t)idx + 1] = dscale.y;
            a.dL_dscale[3 * (size_t)idx + 2] = dscale.z;
        }
        if (a.dL_drot) reinterpret_cast<float4*>(a.dL_drot)[idx] = drot;
    }
    // Resident scratch: the readers clear the rows they have read.  The 64 rows of a wave's Gaussians are 4 KB in a row, so the
    // wave clears them together with four fully coalesced 16-byte stores per lane (rows of Gaussians the view did not see are
    // zero already; writing them again costs nothing extra).  Every lane's loads of its own row have been consumed by now.
    // (Each lane clearing its own row -- four stores of 16 bytes at a 64-byte stride, 64 partial lines per instruction -- cost the
    //  kernel 30 us at config 3, 44 -> 74; right behind the loads, where it also delayed every other input's request, the same.)
    if (a.clear_scratch) {
        const size_t wave_row0 = (size_t)blockIdx.x * 256 + (threadIdx.x & ~63u);
        float4* wbase = reinterpret_cast<float4*>(a.acc + wave_row0 * DGR_ACC_STRIDE);
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int piece = k * 64 + (int)(threadIdx.x & 63u);           // 16-byte piece of the wave's 4 KB
            if (wave_row0 + (size_t)(piece >> 2) < (size_t)a.P) wbase[piece] = z;
        }
    }

    if (a.track_off) {  // no pose gradient asked for: zeros (L/rasterize_points.cu:186)
        if (blockIdx.x == 0 && threadIdx.x < 16) a.dL_dview[threadIdx.x] = 0.0f;
        return;
    }
    __shared__ double red[16][12];
    if (a.det_pose)
        pose_block_reduce_det(pose, a.det_pose, a.ticket, a.dL_dview, red, a.clear_scratch != 0);
    else
        pose_block_reduce(pose, a.pose_part, a.ticket, a.dL_dview, red, a.clear_scratch != 0);
}

// ------------------------------------------------------------------------------------------------
// Batched per-Gaussian backward: the V views of a batch in ONE launch, the gradients of the shared
// Gaussians SUMMED OVER THE VIEWS in registers and written once.  Per view a lane reads that view's 64-byte accumulator row,
// radius, clamp bits and SH direction derivatives and runs the code of the one-view kernel (bwd_view_terms); position,
// covariance, scale and rotation are read once; the covariance backward -- linear in dL_dcov3D -- runs once on the summed
// dL_dcov3D.  A view's terms are formed exactly as the one-view kernel forms them and added in view order with FMA
// contraction off, so -- for the same accumulator rows -- dL_dmeans3D / dL_dsh / dL_dopacity / dL_dcov3D are what accumulating
// the one-view outputs view after view (autograd's `.grad +=`) gives, operation for operation, without the V dense 248-byte
// rows per Gaussian that costs.
// Pose gradients and dL_dmean2D (densification statistics) stay per view.  Light variant only.
__device__ __forceinline__ PreprocessBwdArgs batch_view_args(const PreprocessBwdBatchArgs& b, int v) {
    PreprocessBwdArgs a = b.base;
    const BwdViewPart& p = b.v[v];
    a.view = p.view; a.proj = p.proj; a.campos = p.campos; a.perspec = p.perspec; a.radii = p.radii; a.geom = p.geom;
    a.acc = const_cast<float*>(p.acc); a.dL_dmean2D = p.dL_dmean2D; a.pose_part = p.pose_part; a.ticket = p.ticket; a.dL_dview = p.dL_dview;
    return a;
}
__global__ void __launch_bounds__(256, DGR_BWD_BATCH_WAVES) preprocess_bwd_batch_kernel(PreprocessBwdBatchArgs b) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int P = b.base.P, V = b.V;
    const bool in = idx < P;
    __shared__ float sht[4 * SHT_ROWS * SHT_LD];
    __shared__ double red[DGR_MAX_BATCH_VIEWS][16][12];  // every view's row sums: ONE barrier behind the view loop
    float3 m = make_float3(0.f, 0.f, 0.f), sc_in = make_float3(0.f, 0.f, 0.f);
    float4 q_in = make_float4(0.f, 0.f, 0.f, 0.f);
    float c3[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (in) {
        m = make_float3(b.base.means3D[3 * idx], b.base.means3D[3 * idx + 1], b.base.means3D[3 * idx + 2]);
        if (b.base.cov3D_precomp) {
            const float* c3p = b.base.cov3D_precomp + 6 * (size_t)idx;
#pragma unroll
            for (int i = 0; i < 6; i++) c3[i] = c3p[i];
        } else if (b.base.scales) {
            // the forward's value, re-formed from scale and rotation (same expression, same bits) instead of read back:
            // a view that culled the Gaussian never stored it
            compute_cov3d(b.base.scales, b.base.rotations, b.base.scale_modifier, idx, c3);
        }
        if (b.base.scales) {
            sc_in = make_float3(b.base.scales[3 * idx], b.base.scales[3 * idx + 1], b.base.scales[3 * idx + 2]);
            q_in = make_float4(b.base.rotations[4 * idx], b.base.rotations[4 * idx + 1], b.base.rotations[4 * idx + 2],
                               b.base.rotations[4 * idx + 3]);
        }
    }
    float3 dmean_s = make_float3(0.f, 0.f, 0.f), dcol_s = make_float3(0.f, 0.f, 0.f);
    float dcov_s[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float dop_s = 0.0f;
    float dsh[48];
#pragma unroll
    for (int e = 0; e < 48; e++) dsh[e] = 0.0f;
    bool any_map = false;
#pragma unroll 1
    for (int v = 0; v < V; v++) {
        const PreprocessBwdArgs a = batch_view_args(b, v);
        float pose[12];
#pragma unroll
        for (int i = 0; i < 12; i++) pose[i] = 0.0f;
        if (in) {
            const float4* ap = reinterpret_cast<const float4*>(a.acc + (size_t)idx * DGR_ACC_STRIDE);
            const float4 a0 = ap[0], a1 = ap[1], a2 = ap[2], a3 = ap[3];
            const int rad = a.radii[idx];
            const uint8_t cl_in = a.geom.clamped[idx];
            const float4 shd0 = a.geom.shd[idx], shd1 = a.geom.shd[(size_t)P + idx], shd2 = a.geom.shd[2 * (size_t)P + idx];
            const bool vis = rad > 0;
            float acc[16];
            if (vis) {
                acc[0] = a0.x; acc[1] = a0.y; acc[2] = a0.z; acc[3] = a0.w; acc[4] = a1.x; acc[5] = a1.y; acc[6] = a1.z; acc[7] = a1.w;
                acc[8] = a2.x; acc[9] = a2.y; acc[10] = a2.z; acc[11] = a2.w; acc[12] = a3.x; acc[13] = a3.y; acc[14] = a3.z; acc[15] = a3.w;
            } else {
#pragma unroll
                for (int i = 0; i < 16; i++) acc[i] = 0.0f;
            }
            if (a.dL_dmean2D) {
                a.dL_dmean2D[3 * (size_t)idx + 0] = a.map_off ? 0.0f : acc[4];
                a.dL_dmean2D[3 * (size_t)idx + 1] = a.map_off ? 0.0f : acc[5];
                a.dL_dmean2D[3 * (size_t)idx + 2] = 0.0f;
            }
            float3 dmean, dRGB;
            float dcov[6], coef[16];
            bwd_view_terms(a, false, m, c3, acc, vis, cl_in, shd0, shd1, shd2, dmean, dcov, coef, dRGB, pose);
            any_map |= vis && !a.map_off;
            dop_s += acc[9];
            dcol_s.x += acc[0]; dcol_s.y += acc[1]; dcol_s.z += acc[2];
            dmean_s.x += dmean.x; dmean_s.y += dmean.y; dmean_s.z += dmean.z;
#pragma unroll
            for (int i = 0; i < 6; i++) dcov_s[i] += dcov[i];
            const float ch[3] = {dRGB.x, dRGB.y, dRGB.z};
#pragma unroll
            for (int e = 0; e < 48; e++) dsh[e] += coef[e / 3] * ch[e % 3];  // (contraction is off: product, then sum)
        }
        if (a.track_off) {
            if (blockIdx.x == 0 && threadIdx.x < 16) a.dL_dview[threadIdx.x] = 0.0f;
        } else {
            pose_rows_to_lds(pose, red[v]);  // (no barrier inside the view loop: the waves run through it independently)
        }
    }
    const int M = b.base.M;
    const bool blk_fast = b.base.dL_dsh && b.base.sh_vec_ok && M == 16 && (size_t)blockIdx.x * 256 + 256 <= (size_t)P;
    if (in) {
        float3 dscale = make_float3(0, 0, 0);
        float4 drot = make_float4(0, 0, 0, 0);
        if (any_map && b.base.scales) cov3d_backward_terms(sc_in, q_in, b.base.scale_modifier, dcov_s, dscale, drot);
        if (b.base.dL_dopacity) b.base.dL_dopacity[idx] = dop_s;
        if (b.base.dL_dcolor) {
            b.base.dL_dcolor[3 * (size_t)idx + 0] = dcol_s.x;
            b.base.dL_dcolor[3 * (size_t)idx + 1] = dcol_s.y;
            b.base.dL_dcolor[3 * (size_t)idx + 2] = dcol_s.z;
        }
        if (b.base.dL_dmean3D) {
            b.base.dL_dmean3D[3 * (size_t)idx + 0] = dmean_s.x;
            b.base.dL_dmean3D[3 * (size_t)idx + 1] = dmean_s.y;
            b.base.dL_dmean3D[3 * (size_t)idx + 2] = dmean_s.z;
        }
        if (b.base.dL_dcov3D) {
#pragma unroll
            for (int i = 0; i < 6; i++) b.base.dL_dcov3D[6 * (size_t)idx + i] = dcov_s[i];
        }
        if (b.base.dL_dscale) {
            b.base.dL_dscale[3 * (size_t)idx + 0] = dscale.x;
            b.base.dL_dscale[3 * (size_t)idx + 1] = dscale.y;
            b.base.dL_dscale[3 * (size_t)idx + 2] = dscale.z;
        }
        if (b.base.dL_drot) reinterpret_cast<float4*>(b.base.dL_drot)[idx] = drot;
        if (b.base.dL_dsh && M > 0 && !blk_fast) {
            float* out = b.base.dL_dsh + (size_t)idx * M * 3;
#pragma unroll
            for (int e = 0; e < 48; e++)
                if (e < 3 * M) out[e] = dsh[e];
        }
    }
    if (blk_fast) lanes_to_sh_rows(dsh, b.base.dL_dsh, (size_t)blockIdx.x * 256, sht);
    if (b.base.track_off) return;
    // Pose gradients of all views: one barrier, then wave 0 adds every view's partial, waits once, draws every view's ticket
    // (issued back to back: one L2 round trip for the batch) and finishes the views whose last block this is.
    __syncthreads();
    if (threadIdx.x >= 64) return;
    if (b.v[0].det_pose) {
        // deterministic gradients: every view's partial of this block STORED to the view's own [blocks, 12] array, the views'
        // tickets drawn together, and whoever drew a view's last one adds its rows in a fixed order (pose_block_reduce_det)
        __shared__ double lane_sum[64][12];
#pragma unroll 1
        for (int v = 0; v < V; v++)
            if (threadIdx.x < 12) {
                double part = 0.0;
#pragma unroll
                for (int r = 0; r < 16; r++) part += red[v][r][threadIdx.x];
                __hip_atomic_store(b.v[v].det_pose + (size_t)blockIdx.x * 12 + threadIdx.x, part, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        uint32_t td = 0u;
#pragma unroll 1
        for (int v = 0; v < V; v++)
            if ((int)threadIdx.x == v) td = __hip_atomic_fetch_add(b.v[v].ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll 1
        for (int v = 0; v < V; v++)
            if ((uint32_t)__builtin_amdgcn_readlane((int)td, v) == gridDim.x - 1) pose_finish_det(b.v[v].det_pose, b.v[v].ticket, b.v[v].dL_dview, lane_sum, false);
        return;
    }
#pragma unroll 1
    for (int v = 0; v < V; v++) pose_add_partial(red[v], b.v[v].pose_part);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    uint32_t t = 0u;
#pragma unroll 1
    for (int v = 0; v < V; v++)
        if ((int)threadIdx.x == v) t = __hip_atomic_fetch_add(b.v[v].ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll 1
    for (int v = 0; v < V; v++)
        pose_finish_if_last((uint32_t)__builtin_amdgcn_readlane((int)t, v), b.v[v].pose_part, b.v[v].dL_dview);
}

// ------------------------------------------------------------------------------------------------
// View-independent half of the per-Gaussian work, shared by the views of a batch: the 3D
// covariance depends on scale and rotation only.  cov3d_fwd is computeCov3D (forward.cu:118-152) exactly as
// preprocess_fwd evaluates it (same expression: the six floats are bit-identical), to be passed to every view as
// `cov3D_precomp`; cov3d_bwd is its backward (L/cuda_rasterizer/backward.cu:280-343), which is LINEAR in dL_dcov3D, so
// the views' dL_dcov3D are summed first (autograd does that) and converted to dL_dscale / dL_drot once.
__global__ void __launch_bounds__(256) cov3d_fwd_kernel(int P, const float* __restrict__ scales, const float* __restrict__ rotations,
                                                        float mod, float* __restrict__ cov3D) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= P) return;
    const float3 sc = make_float3(scales[3 * idx], scales[3 * idx + 1], scales[3 * idx + 2]);
    const float4 q = make_float4(rotations[4 * idx], rotations[4 * idx + 1], rotations[4 * idx + 2], rotations[4 * idx + 3]);
    M3 R;
    quat_to_R(q, R);
    const M3 S = diag3(mod * sc.x, mod * sc.y, mod * sc.z);
    const M3 Mm = mul(S, R);
    const M3 Sigma = mul(transpose(Mm), Mm);
    float* o = cov3D + 6 * (size_t)idx;
    o[0] = Sigma.m[0][0]; o[1] = Sigma.m[0][1]; o[2] = Sigma.m[0][2];
    o[3] = Sigma.m[1][1]; o[4] = Sigma.m[1][2]; o[5] = Sigma.m[2][2];
}
__global__ void __launch_bounds__(256) cov3d_bwd_kernel(int P, const float* __restrict__ scales, const float* __restrict__ rotations,
                                                        float mod, const float* __restrict__ dL_dcov3D, float* __restrict__ dL_dscale,
                                                        float* __restrict__ dL_drot) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= P) return;
    const float* dcov = dL_dcov3D + 6 * (size_t)idx;
    const float3 sc = make_float3(scales[3 * idx], scales[3 * idx + 1], scales[3 * idx + 2]);
    const float4 q = make_float4(rotations[4 * idx], rotations[4 * idx + 1], rotations[4 * idx + 2], rotations[4 * idx + 3]);
    const float r = q.x, x = q.y, y = q.z, z = q.w;
    M3 R;
    quat_to_R(q, R);
    const float3 s = make_float3(mod * sc.x, mod * sc.y, mod * sc.z);
    const M3 Mm = mul(diag3(s.x, s.y, s.z), R);
    M3 dSigma;
    dSigma.m[0][0] = dcov[0]; dSigma.m[0][1] = 0.5f * dcov[1]; dSigma.m[0][2] = 0.5f * dcov[2];
    dSigma.m[1][0] = 0.5f * dcov[1]; dSigma.m[1][1] = dcov[3]; dSigma.m[1][2] = 0.5f * dcov[4];
    dSigma.m[2][0] = 0.5f * dcov[2]; dSigma.m[2][1] = 0.5f * dcov[4]; dSigma.m[2][2] = dcov[5];
    M3 M2;
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int rr = 0; rr < 3; rr++) M2.m[c][rr] = Mm.m[c][rr] * 2.0f;
    const M3 dL_dM = mul(M2, dSigma);
    const M3 Rt = transpose(R);
    M3 dMt = transpose(dL_dM);
    float3 dscale;
    dscale.x = dot3(make_float3(Rt.m[0][0], Rt.m[0][1], Rt.m[0][2]), make_float3(dMt.m[0][0], dMt.m[0][1], dMt.m[0][2]));
    dscale.y = dot3(make_float3(Rt.m[1][0], Rt.m[1][1], Rt.m[1][2]), make_float3(dMt.m[1][0], dMt.m[1][1], dMt.m[1][2]));
    dscale.z = dot3(make_float3(Rt.m[2][0], Rt.m[2][1], Rt.m[2][2]), make_float3(dMt.m[2][0], dMt.m[2][1], dMt.m[2][2]));
#pragma unroll
    for (int k = 0; k < 3; k++) { dMt.m[0][k] *= s.x; dMt.m[1][k] *= s.y; dMt.m[2][k] *= s.z; }
    float4 drot;
    drot.x = 2 * z * (dMt.m[0][1] - dMt.m[1][0]) + 2 * y * (dMt.m[2][0] - dMt.m[0][2]) + 2 * x * (dMt.m[1][2] - dMt.m[2][1]);
    drot.y = 2 * y * (dMt.m[1][0] + dMt.m[0][1]) + 2 * z * (dMt.m[2][0] + dMt.m[0][2]) + 2 * r * (dMt.m[1][2] - dMt.m[2][1]) - 4 * x * (dMt.m[2][2] + dMt.m[1][1]);
    drot.z = 2 * x * (dMt.m[1][0] + dMt.m[0][1]) + 2 * r * (dMt.m[2][0] - dMt.m[0][2]) + 2 * z * (dMt.m[1][2] + dMt.m[2][1]) - 4 * y * (dMt.m[2][2] + dMt.m[0][0]);
    drot.w = 2 * r * (dMt.m[0][1] - dMt.m[1][0]) + 2 * x * (dMt.m[2][0] + dMt.m[0][2]) + 2 * y * (dMt.m[1][2] + dMt.m[2][1]) - 4 * z * (dMt.m[1][1] + dMt.m[0][0]);
    // the reference scales dL_dscale by the modifier through `s` only (backward.cu:318-322): so does preprocess_bwd
    dL_dscale[3 * (size_t)idx + 0] = dscale.x;
    dL_dscale[3 * (size_t)idx + 1] = dscale.y;
    dL_dscale[3 * (size_t)idx + 2] = dscale.z;
    reinterpret_cast<float4*>(dL_drot)[idx] = drot;
}

}  // namespace dgr

namespace dgr {
hipError_t launch_cov3d_forward(int P, const float* scales, const float* rotations, float mod, float* cov3D, hipStream_t stream) {
    if (P <= 0) return hipSuccess;
    launch(cov3d_fwd_kernel, dim3((P + 255) / 256), dim3(256), stream, P, scales, rotations, mod, cov3D);
    return hipGetLastError();
}
hipError_t launch_cov3d_backward(int P, const float* scales, const float* rotations, float mod, const float* dL_dcov3D,
                                 float* dL_dscale, float* dL_drot, hipStream_t stream) {
    if (P <= 0) return hipSuccess;
    launch(cov3d_bwd_kernel, dim3((P + 255) / 256), dim3(256), stream, P, scales, rotations, mod, dL_dcov3D, dL_dscale, dL_drot);
    return hipGetLastError();
}
namespace {
__global__ void __launch_bounds__(256) zero_fill_kernel(float4* dst, size_t n16) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) dst[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}
}  // namespace
hipError_t launch_zero_fill(void* dst, size_t bytes, hipStream_t stream) {
    const size_t n16 = bytes / 16;
    if (n16 == 0) return hipSuccess;
    const unsigned blocks = (unsigned)std::min<size_t>((n16 + 255) / 256, 256 * 16);
    launch(zero_fill_kernel, dim3(blocks), dim3(256), stream, (float4*)dst, n16);
    return hipGetLastError();
}
hipError_t launch_preprocess_fwd(const PreprocessFwdArgs& a, hipStream_t stream) {
    if (a.P <= 0) return hipSuccess;
    launch(preprocess_fwd_kernel, dim3((a.P + 255) / 256), dim3(256), stream, a);
    return hipGetLastError();
}
hipError_t launch_preprocess_fwd_batch(const PreprocessFwdBatchArgs& b, hipStream_t stream) {
    if (b.base.P <= 0 || b.V <= 0) return hipSuccess;
    launch(preprocess_fwd_batch_kernel, dim3((b.base.P + 255) / 256), dim3(256), stream, b);
    return hipGetLastError();
}
hipError_t launch_preprocess_bwd(const PreprocessBwdArgs& a, hipStream_t stream) {
    const int blocks = (a.P + 255) / 256;
    if (blocks <= 0) return hipSuccess;
    launch(preprocess_bwd_kernel, dim3(blocks), dim3(256), stream, a);
    return hipGetLastError();
}
hipError_t launch_preprocess_bwd_batch(const PreprocessBwdBatchArgs& b, hipStream_t stream) {
    if (b.base.P <= 0 || b.V <= 0) return hipSuccess;
    launch(preprocess_bwd_batch_kernel, dim3((b.base.P + 255) / 256), dim3(256), stream, b);
    return hipGetLastError();
}
hipError_t launch_mark_visible(int P, const float* means, const float* view, uint8_t* present, hipStream_t stream) {
    if (P <= 0) return hipSuccess;
    launch(mark_visible_kernel, dim3((P + 255) / 256), dim3(256), stream, P, means, view, present);
    return hipGetLastError();
}
}  // namespace dgr
