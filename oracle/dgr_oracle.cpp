// dgr_oracle.cpp -- CPU restatement of the reference's rasterization path.
//
// TEST INFRASTRUCTURE ONLY.  Nothing in the product path (diff-gaussian-rasterization_amd/)
// may import, link or call this file; only tests/, __graft_entry__.smoke() and the
// cpu_baseline leg of bench.py do, and there only as the checker / reported baseline.
//
// Pinning status: the reference ships no tests or golden vectors and cannot
// be built in this image without stand-in CUDA/CUB headers, so it is NOT executed here.
// This restatement is pinned to the known-answer table of SURVEY.md Appendix C (values
// recorded from a CPU execution of the reference's own sources during the survey):
// visible counts, sha256 of radii, R, sum n_contrib, image sums and dL_dview matrices.
// Beyond those known answers parity is unpinned -- see DESIGN.md "Oracle".
//
// Arithmetic follows the reference operation by operation (same association order, no
// FMA contraction: build with -O2 -ffp-contract=off, no -march) so that threshold
// decisions (power>0, alpha<1/17, T<1e-4, ceil(3 sqrt(l)), det==0, z<=0.2) agree.
// Files restated (L = diff-gaussian-rasterization-light, F = ...-full, cr = cuda_rasterizer):
//   L/cr/auxiliary.h, L/cr/forward.cu, L/cr/backward.cu, L/cr/rasterizer_impl.cu,
//   L/rasterize_points.cu, and the F/ counterparts where the variants differ.
//
// Float atomics of the reference (per-Gaussian gradient scatter, gau_uncertainty) are
// order-nondeterministic there; here every such sum is accumulated in double and rounded
// once, i.e. the centre of the reference's own run-to-run spread.

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <map>
#include <numeric>
#include <string>
#include <vector>

namespace {

// The reference calls exp/sqrt/ceil unqualified.  Under nvcc those bind to the float
// overloads (default here).  -DDGRO_C_MATH=1 binds them to the C double functions instead,
// which is what the survey's CPU execution did; that build exists
// only so tests can reproduce Appendix C's digits.
// ---- expf, restated.  The blend loops' exp decides hard thresholds (alpha >= 15/255, T < 1e-4, the median's T > 0.5) and its
// last bit is amplified by the light backward (T_final = 1 - alpha image, division by 1 - alpha), so "the oracle's bits" must
// not depend on which C library -- or which ifunc variant of it -- the box that runs the tests happens to have.  This is the
// algorithm glibc >= 2.27 uses for expf (ARM optimized routines: N = 32 table entries of 2^(i/N), a cubic in double, ~0.502
// ulp), written out in IEEE double operations whose results are the same on every machine: z = x N / ln 2, k = round(z) by the
// 1.5 * 2^52 shift, r = z - k, s = 2^(k/N) assembled from the table, y = 1 + r (C2 + r (C1 + r C0)) by Horner's rule with
// fused multiply-adds (std::fma is correctly rounded whether or not the CPU has the instruction), result (float)(y s).
// The HIP kernels evaluate exactly this sequence (csrc/exact_math.h: exp_ref; tests/test_hip_exact_math.py compares the two
// bit for bit); tests/test_oracle_expf.py pins this function with known-answer vectors and compares it with the host's expf
// where that is glibc >= 2.27 (glibc sums the cubic as (C0 r + C1) r^2 + (C2 r + 1): the two agree on all but about one
// argument in 2^28).  Range handling as glibc's: NaN, overflow above 88.72, 0 below -103.97.
const uint64_t EXP2F_TAB[32] = {
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull, 0x3fef72b83c7d517bull,
    0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull, 0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull,
    0x3feedea64c123422ull, 0x3feece086061892dull, 0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull,
    0x3feea47eb03a5585ull, 0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
    0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull, 0x3feee89f995ad3adull,
    0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull, 0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full,
    0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};
inline float expf_restated(float x) {
    if (x != x) return x;
    if (x > 0x1.62e42ep6f) return std::numeric_limits<float>::infinity();
    if (x < -0x1.9fe368p6f) return 0.0f;
    constexpr double INVLN2N = 0x1.71547652b82fep+0 * 32.0, SHIFT = 0x1.8p+52;
    constexpr double C0 = 0x1.c6af84b912394p-5 / 32 / 32 / 32, C1 = 0x1.ebfce50fac4f3p-3 / 32 / 32, C2 = 0x1.62e42ff0c52d6p-1 / 32;
    const double z = INVLN2N * (double)x;
    double kd = z + SHIFT;  // round to nearest even; the integer k sits in the low mantissa bits
    uint64_t ki;
    std::memcpy(&ki, &kd, 8);
    kd -= SHIFT;
    const double r = z - kd;  // in [-1/2, 1/2]
    const uint64_t t = EXP2F_TAB[ki & 31u] + (ki << 47);
    double s;
    std::memcpy(&s, &t, 8);   // 2^(k/32)
    double p = std::fma(r, C0, C1);
    p = std::fma(p, r, C2);
    const double y = std::fma(p, r, 1.0);
    return (float)(y * s);
}

// ---- expf, restated a second way: fp32 only (round 8; the default of the float build).  Any expf good to <= 2 ulp is as
// faithful to the reference's `exp` (nvcc's expf on a float argument, L/cr/forward.cu:364, backward.cu:565) as glibc's; this one is
// made of operations a GPU's single-precision pipe issues at full rate, and csrc/exact_math.h: exp_p32 evaluates exactly this
// sequence (tests/test_hip_exact_math.py: bit for bit).  Every step is one IEEE single-precision operation -- std::fmaf is
// correctly rounded whether or not the CPU has the instruction -- or an integer shift / add, so the bits are the same on
// every machine:
//   t = fma(x, log2 e, M), M = 1.5 * 2^23 + 64: the round-to-nearest-even integer k of x log2 e sits in t's low mantissa bits,
//   biased by 64;  k = t - M;  r = x - k ln 2 in two fused steps (ln 2 = hi + lo);  p = 1 + r + r^2 / 2 + r^3 (C3 + r (C4 + r (C5
//   + r C6))) by Horner's rule, the four free coefficients minimising the relative error on |r| <= 0.3467 (3.6e-9);
//   2^(k + 64) p by adding (bits of t) << 23 to the bits of p (M's bits vanish in the shift);  times 2^-64 -- exact for normal
//   results, one rounding for denormal ones.
// Error against exp() in double over all 1 120 927 745 floats of [-104, -0]: <= 0.892 ulp (0.858 ulp where the result is denormal),
// 99.52 % correctly rounded (dgro_expf_p32_scan, tests/test_oracle_expf.py).  Arguments below -104 (and NaN) are evaluated at
// -104, where the result is 0 as at every smaller argument.  Supported: x <= 0 (the blend loops reject power > 0 before).
inline float expf_p32(float x) {
    x = std::fmax(x, -104.0f);  // (fmaxf: a NaN argument yields -104, as v_max_f32 does)
    constexpr float LOG2E = 0x1.715476p+0f, M = 12582976.0f, NLN2HI = -0x1.62e430p-1f, NLN2LO = 0x1.05c610p-29f;
    constexpr float C3 = 0x1.5554a4p-3f, C4 = 0x1.555688p-5f, C5 = 0x1.122faep-7f, C6 = 0x1.6b6e26p-10f;
    const float t = std::fma(x, LOG2E, M);
    const float k = t - M;
    float r = std::fma(k, NLN2HI, x);
    r = std::fma(k, NLN2LO, r);
    float p = std::fma(C6, r, C5);
    p = std::fma(p, r, C4);
    p = std::fma(p, r, C3);
    p = std::fma(p, r, 0.5f);
    p = std::fma(p, r, 1.0f);
    p = std::fma(p, r, 1.0f);
    uint32_t tb, pb;
    std::memcpy(&tb, &t, 4);
    std::memcpy(&pb, &p, 4);
    const uint32_t b = (tb << 23) + pb;
    float y;
    std::memcpy(&y, &b, 4);
    return y * 0x1p-64f;
}
// which of the two restated expf the float build's blend loops call: 0 = expf_p32 (default; the HIP kernels' alpha mode 0),
// 1 = expf_restated, glibc's algorithm (the kernels' alpha mode 2; rounds 5-7's default).  dgro_set_exp_mode.
int g_exp_mode = 0;

#if defined(DGRO_C_MATH) && DGRO_C_MATH
inline double m_exp(float x) { return ::exp((double)x); }
inline double m_sqrt(float x) { return ::sqrt((double)x); }
inline double m_ceil(double x) { return ::ceil(x); }
#else
inline float m_exp(float x) { return g_exp_mode ? expf_restated(x) : expf_p32(x); }
inline float m_sqrt(float x) { return std::sqrt(x); }
inline float m_ceil(float x) { return std::ceil(x); }
#endif

constexpr int BLOCK_X = 16, BLOCK_Y = 16;  // L/cr/config.h:16-17

// --- small column-major 3x3 / vec3 helpers with the operation order of the vendored
// GLM 0.9.9.9 (third_party/glm/glm/detail/type_mat3x3.inl:486-520, func_geometric.inl:48-54):
// m[c][r] is column c, row r; (A*B)[c][r] = A[0][r]*B[c][0] + A[1][r]*B[c][1] + A[2][r]*B[c][2].
struct V3 {
    float x, y, z;
};
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(float s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator/(V3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline V3& operator+=(V3& a, V3 b) {
    a.x += b.x; a.y += b.y; a.z += b.z;
    return a;
}
inline float dot(V3 a, V3 b) {
    float tx = a.x * b.x, ty = a.y * b.y, tz = a.z * b.z;
    return tx + ty + tz;
}
inline float length(V3 a) { return std::sqrt(dot(a, a)); }  // glm::length: always the float overload

struct M3 {
    float m[3][3];
    float* operator[](int c) { return m[c]; }
    const float* operator[](int c) const { return m[c]; }
};
inline M3 mat3(float a0, float a1, float a2, float b0, float b1, float b2, float c0, float c1,
               float c2) {
    M3 r;
    r[0][0] = a0; r[0][1] = a1; r[0][2] = a2;
    r[1][0] = b0; r[1][1] = b1; r[1][2] = b2;
    r[2][0] = c0; r[2][1] = c1; r[2][2] = c2;
    return r;
}
inline M3 operator*(const M3& A, const M3& B) {
    M3 R;
    for (int c = 0; c < 3; c++)
        for (int r = 0; r < 3; r++) R[c][r] = A[0][r] * B[c][0] + A[1][r] * B[c][1] + A[2][r] * B[c][2];
    return R;
}
inline M3 operator*(float s, const M3& A) {
    M3 R;
    for (int c = 0; c < 3; c++)
        for (int r = 0; r < 3; r++) R[c][r] = A[c][r] * s;  // GLM: m[c] * s
    return R;
}
inline M3 transpose(const M3& A) {
    M3 R;
    for (int c = 0; c < 3; c++)
        for (int r = 0; r < 3; r++) R[c][r] = A[r][c];
    return R;
}
inline V3 col(const M3& A, int c) { return {A[c][0], A[c][1], A[c][2]}; }

// L/cr/auxiliary.h:22-39
const float SH_C0 = 0.28209479177387814f;
const float SH_C1 = 0.4886025119029199f;
const float SH_C2[] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                       -1.0925484305920792f, 0.5462742152960396f};
const float SH_C3[] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                       0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                       -0.5900435899266435f};

// L/cr/auxiliary.h:41-44 (double arithmetic, narrowed on return)
inline float ndc2Pix(float v, int S) { return ((v + 1.0) * S - 1.0) * 0.5; }

// L/cr/auxiliary.h:46-56
inline void getRect(float px, float py, int max_radius, uint32_t gx, uint32_t gy, uint32_t* rmin,
                    uint32_t* rmax) {
    rmin[0] = std::min<uint32_t>(gx, (uint32_t)std::max(0, (int)((px - max_radius) / BLOCK_X)));
    rmin[1] = std::min<uint32_t>(gy, (uint32_t)std::max(0, (int)((py - max_radius) / BLOCK_Y)));
    rmax[0] = std::min<uint32_t>(gx, (uint32_t)std::max(0, (int)((px + max_radius + BLOCK_X - 1) / BLOCK_X)));
    rmax[1] = std::min<uint32_t>(gy, (uint32_t)std::max(0, (int)((py + max_radius + BLOCK_Y - 1) / BLOCK_Y)));
}

// L/cr/auxiliary.h:58-77
inline V3 transformPoint4x3(V3 p, const float* m) {
    return {m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12], m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
            m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14]};
}
struct V4 {
    float x, y, z, w;
};
inline V4 transformPoint4x4(V3 p, const float* m) {
    return {m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12], m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
            m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14], m[3] * p.x + m[7] * p.y + m[11] * p.z + m[15]};
}
// L/cr/auxiliary.h:91-99
inline V3 transformVec4x3Transpose(V3 p, const float* m) {
    return {m[0] * p.x + m[1] * p.y + m[2] * p.z, m[4] * p.x + m[5] * p.y + m[6] * p.z,
            m[8] * p.x + m[9] * p.y + m[10] * p.z};
}
// L/cr/auxiliary.h:109-119
inline V3 dnormvdv(V3 v, V3 dv) {
    float sum2 = v.x * v.x + v.y * v.y + v.z * v.z;
    float invsum32 = 1.0f / m_sqrt(sum2 * sum2 * sum2);
    V3 r;
    r.x = ((+sum2 - v.x * v.x) * dv.x - v.y * v.x * dv.y - v.z * v.x * dv.z) * invsum32;
    r.y = (-v.x * v.y * dv.x + (sum2 - v.y * v.y) * dv.y - v.z * v.y * dv.z) * invsum32;
    r.z = (-v.x * v.z * dv.x - v.y * v.z * dv.y + (sum2 - v.z * v.z) * dv.z) * invsum32;
    return r;
}

// L/cr/rasterizer_impl.cu:35-50
uint32_t getHigherMsb(uint32_t n) {
    uint32_t msb = sizeof(n) * 4;
    uint32_t step = msb;
    while (step > 1) {
        step /= 2;
        if (n >> msb) msb += step;
        else msb -= step;
    }
    if (n >> msb) msb++;
    return msb;
}

struct State {
    int P = 0, W = 0, H = 0, R = 0;
    uint32_t gx = 0, gy = 0;
    // GeometryState (L/cr/rasterizer_impl.h:29-44)
    std::vector<float> depths, means2D, cov3D, conic_opacity, rgb;
    std::vector<uint8_t> clamped;
    std::vector<int32_t> radii;
    std::vector<uint32_t> tiles_touched, point_offsets;
    // BinningState (L/cr/rasterizer_impl.h:54-64)
    std::vector<uint64_t> keys_unsorted, keys;
    std::vector<uint32_t> point_list_unsorted, point_list;
    // ImageState (L/cr/rasterizer_impl.h:46-52; F adds accum_alpha, n_valid_contrib)
    std::vector<uint32_t> ranges, n_contrib, n_valid_contrib;
    std::vector<float> final_T;
    // light backward scratch (L/rasterize_points.cu:185-187)
    std::vector<float> dgndcs_dview, dg_camd;
};

// ---------------------------------------------------------------- forward preprocess
// */cr/forward.cu:20-71
V3 colorFromSH(int idx, int deg, int M, const float* means, V3 campos, const float* shs, uint8_t* clamped) {
    V3 pos = {means[3 * idx], means[3 * idx + 1], means[3 * idx + 2]};
    V3 dir = pos - campos;
    dir = dir / length(dir);
    const V3* sh = reinterpret_cast<const V3*>(shs) + (size_t)idx * M;
    V3 result = SH_C0 * sh[0];
    if (deg > 0) {
        float x = dir.x, y = dir.y, z = dir.z;
        result = result - SH_C1 * y * sh[1] + SH_C1 * z * sh[2] - SH_C1 * x * sh[3];
        if (deg > 1) {
            float xx = x * x, yy = y * y, zz = z * z;
            float xy = x * y, yz = y * z, xz = x * z;
            result = result + SH_C2[0] * xy * sh[4] + SH_C2[1] * yz * sh[5] +
                     SH_C2[2] * (2.0f * zz - xx - yy) * sh[6] + SH_C2[3] * xz * sh[7] +
                     SH_C2[4] * (xx - yy) * sh[8];
            if (deg > 2) {
                result = result + SH_C3[0] * y * (3.0f * xx - yy) * sh[9] + SH_C3[1] * xy * z * sh[10] +
                         SH_C3[2] * y * (4.0f * zz - xx - yy) * sh[11] +
                         SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * sh[12] +
                         SH_C3[4] * x * (4.0f * zz - xx - yy) * sh[13] + SH_C3[5] * z * (xx - yy) * sh[14] +
                         SH_C3[6] * x * (xx - 3.0f * yy) * sh[15];
            }
        }
    }
    result.x += 0.5f; result.y += 0.5f; result.z += 0.5f;
    clamped[3 * idx + 0] = (result.x < 0);
    clamped[3 * idx + 1] = (result.y < 0);
    clamped[3 * idx + 2] = (result.z < 0);
    return {std::max(result.x, 0.0f), std::max(result.y, 0.0f), std::max(result.z, 0.0f)};
}

struct Cov2DFwd {
    V3 t;          // clamped camera-space mean
    float txtz, tytz;
    M3 J, Wm, T, Vrk, cov;
};
// */cr/forward.cu:74-113 (also the recomputation at L/cr/backward.cu:166-199)
inline void cov2DCommon(V3 mean, float fx, float fy, float tanx, float tany, const float* c3, const float* v,
                        Cov2DFwd& o) {
    V3 t = transformPoint4x3(mean, v);
    const float limx = 1.3f * tanx, limy = 1.3f * tany;
    o.txtz = t.x / t.z;
    o.tytz = t.y / t.z;
    t.x = std::min(limx, std::max(-limx, o.txtz)) * t.z;
    t.y = std::min(limy, std::max(-limy, o.tytz)) * t.z;
    o.t = t;
    o.J = mat3(fx / t.z, 0.0f, -(fx * t.x) / (t.z * t.z), 0.0f, fy / t.z, -(fy * t.y) / (t.z * t.z), 0, 0, 0);
    o.Wm = mat3(v[0], v[4], v[8], v[1], v[5], v[9], v[2], v[6], v[10]);
    o.T = o.Wm * o.J;
    o.Vrk = mat3(c3[0], c3[1], c3[2], c3[1], c3[3], c3[4], c3[2], c3[4], c3[5]);
    o.cov = transpose(o.T) * transpose(o.Vrk) * o.T;
}

// */cr/forward.cu:118-152
inline void computeCov3D(V3 scale, float mod, const float* rot, float* cov3D) {
    M3 S = mat3(1, 0, 0, 0, 1, 0, 0, 0, 1);
    S[0][0] = mod * scale.x;
    S[1][1] = mod * scale.y;
    S[2][2] = mod * scale.z;
    float r = rot[0], x = rot[1], y = rot[2], z = rot[3];  // NOT normalised (forward.cu:127)
    M3 R = mat3(1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
                2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
                2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y));
    M3 Mm = S * R;
    M3 Sigma = transpose(Mm) * Mm;
    cov3D[0] = Sigma[0][0]; cov3D[1] = Sigma[0][1]; cov3D[2] = Sigma[0][2];
    cov3D[3] = Sigma[1][1]; cov3D[4] = Sigma[1][2]; cov3D[5] = Sigma[2][2];
}

// */cr/forward.cu:155-256.  Returns -1 when `prefiltered` is violated (device __trap there).
int preprocessForward(State& st, int P, int D, int M, const float* means, const float* scales, float mod,
                      const float* rots, const float* opac, const float* shs, const float* cov3D_pre,
                      const float* colors_pre, const float* view, const float* proj, const float* campos,
                      int W, int H, float tanx, float tany, float fx, float fy, int prefiltered) {
    st.P = P; st.W = W; st.H = H;
    st.gx = (W + BLOCK_X - 1) / BLOCK_X;
    st.gy = (H + BLOCK_Y - 1) / BLOCK_Y;
    st.depths.assign(P, 0.f); st.means2D.assign(2 * (size_t)P, 0.f); st.cov3D.assign(6 * (size_t)P, 0.f);
    st.conic_opacity.assign(4 * (size_t)P, 0.f); st.rgb.assign(3 * (size_t)P, 0.f);
    st.clamped.assign(3 * (size_t)P, 0); st.radii.assign(P, 0); st.tiles_touched.assign(P, 0);
    int bad = 0;
    V3 cam = {0, 0, 0};
    if (campos) cam = {campos[0], campos[1], campos[2]};
#pragma omp parallel for schedule(static)
    for (int idx = 0; idx < P; idx++) {
        V3 p_orig = {means[3 * idx], means[3 * idx + 1], means[3 * idx + 2]};
        // in_frustum, cr/auxiliary.h:139-164
        V4 p_hom = transformPoint4x4(p_orig, proj);
        float p_w = 1.0f / (p_hom.w + 0.0000001f);
        V3 p_proj = {p_hom.x * p_w, p_hom.y * p_w, p_hom.z * p_w};
        V3 p_view = transformPoint4x3(p_orig, view);
        if (p_view.z <= 0.2f) {
            if (prefiltered) {
#pragma omp atomic write
                bad = 1;
            }
            continue;
        }
        const float* c3;
        if (cov3D_pre) c3 = cov3D_pre + 6 * (size_t)idx;
        else {
            computeCov3D({scales[3 * idx], scales[3 * idx + 1], scales[3 * idx + 2]}, mod, rots + 4 * (size_t)idx,
                         &st.cov3D[6 * (size_t)idx]);
            c3 = &st.cov3D[6 * (size_t)idx];
        }
        Cov2DFwd c;
        cov2DCommon(p_orig, fx, fy, tanx, tany, c3, view, c);
        c.cov[0][0] += 0.3f;
        c.cov[1][1] += 0.3f;
        V3 cov = {c.cov[0][0], c.cov[0][1], c.cov[1][1]};
        float det = (cov.x * cov.z - cov.y * cov.y);
        if (det == 0.0f) continue;
        float det_inv = 1.f / det;
        V3 conic = {cov.z * det_inv, -cov.y * det_inv, cov.x * det_inv};
        float mid = 0.5f * (cov.x + cov.z);
        float lambda1 = mid + m_sqrt(std::max(0.1f, mid * mid - det));
        float lambda2 = mid - m_sqrt(std::max(0.1f, mid * mid - det));
        float my_radius = m_ceil(3.f * m_sqrt(std::max(lambda1, lambda2)));
        float pix = ndc2Pix(p_proj.x, W), piy = ndc2Pix(p_proj.y, H);
        uint32_t rmin[2], rmax[2];
        getRect(pix, piy, (int)my_radius, st.gx, st.gy, rmin, rmax);
        if ((rmax[0] - rmin[0]) * (rmax[1] - rmin[1]) == 0) continue;
        if (!colors_pre) {
            V3 res = colorFromSH(idx, D, M, means, cam, shs, st.clamped.data());
            st.rgb[3 * (size_t)idx + 0] = res.x;
            st.rgb[3 * (size_t)idx + 1] = res.y;
            st.rgb[3 * (size_t)idx + 2] = res.z;
        }
        st.depths[idx] = p_view.z;
        st.radii[idx] = (int)my_radius;
        st.means2D[2 * (size_t)idx] = pix;
        st.means2D[2 * (size_t)idx + 1] = piy;
        st.conic_opacity[4 * (size_t)idx + 0] = conic.x;
        st.conic_opacity[4 * (size_t)idx + 1] = conic.y;
        st.conic_opacity[4 * (size_t)idx + 2] = conic.z;
        st.conic_opacity[4 * (size_t)idx + 3] = opac[idx];
        st.tiles_touched[idx] = (rmax[1] - rmin[1]) * (rmax[0] - rmin[0]);
    }
    return bad ? -1 : 0;
}

// ---------------------------------------------------------------- keys / sort / ranges
// L/cr/rasterizer_impl.cu:70-138, 283-323 (scan, duplicateWithKeys, SortPairs, identifyTileRanges).
// CUB's contract restated: inclusive sum; STABLE LSD radix sort on key bits [0, 32+bit).
void binning(State& st) {
    const int P = st.P;
    st.point_offsets.resize(P);
    uint32_t run = 0;
    for (int i = 0; i < P; i++) {
        run += st.tiles_touched[i];
        st.point_offsets[i] = run;
    }
    st.R = P ? (int)st.point_offsets[P - 1] : 0;
    const size_t R = st.R;
    st.keys_unsorted.assign(R, 0); st.point_list_unsorted.assign(R, 0);
#pragma omp parallel for schedule(static)
    for (int idx = 0; idx < P; idx++) {
        if (st.radii[idx] > 0) {
            uint32_t off = (idx == 0) ? 0 : st.point_offsets[idx - 1];
            uint32_t rmin[2], rmax[2];
            getRect(st.means2D[2 * (size_t)idx], st.means2D[2 * (size_t)idx + 1], st.radii[idx], st.gx, st.gy, rmin, rmax);
            uint32_t dbits;
            std::memcpy(&dbits, &st.depths[idx], 4);
            for (int y = rmin[1]; y < (int)rmax[1]; y++)
                for (int x = rmin[0]; x < (int)rmax[0]; x++) {
                    uint64_t key = (uint64_t)(y * st.gx + x);
                    key <<= 32;
                    key |= dbits;
                    st.keys_unsorted[off] = key;
                    st.point_list_unsorted[off] = idx;
                    off++;
                }
        }
    }
    const int bit = getHigherMsb(st.gx * st.gy);
    const int end_bit = 32 + bit;
    const uint64_t mask = end_bit >= 64 ? ~0ull : ((1ull << end_bit) - 1);
    std::vector<uint32_t> order(R);
    std::iota(order.begin(), order.end(), 0u);
    const uint64_t* ku = st.keys_unsorted.data();
    std::stable_sort(order.begin(), order.end(),
                     [&](uint32_t a, uint32_t b) { return (ku[a] & mask) < (ku[b] & mask); });
    st.keys.resize(R); st.point_list.resize(R);
    for (size_t i = 0; i < R; i++) {
        st.keys[i] = ku[order[i]];
        st.point_list[i] = st.point_list_unsorted[order[i]];
    }
    const size_t tiles = (size_t)st.gx * st.gy;
    st.ranges.assign(2 * tiles, 0);
    for (size_t idx = 0; idx < R; idx++) {
        uint32_t curr = (uint32_t)(st.keys[idx] >> 32);
        if (idx == 0) st.ranges[2 * curr] = 0;
        else {
            uint32_t prev = (uint32_t)(st.keys[idx - 1] >> 32);
            if (curr != prev) {
                st.ranges[2 * prev + 1] = (uint32_t)idx;
                st.ranges[2 * curr] = (uint32_t)idx;
            }
        }
        if (idx == R - 1) st.ranges[2 * curr + 1] = (uint32_t)R;
    }
}

// ---------------------------------------------------------------- forward blend
// light: L/cr/forward.cu:261-412;  full: F/cr/forward.cu:261-396.
template <bool LIGHT>
void renderForward(State& st, const float* features, const float* bg, const float* gt_depth, float* out_color,
                   float* out_depth, float* out_median, float* out_alpha_or_unc, float* out_depth_var,
                   float* gau_unc, int32_t* gau_px) {
    const int W = st.W, H = st.H;
    const size_t N = (size_t)W * H;
    st.n_contrib.assign(N, 0);
    if (!LIGHT) { st.final_T.assign(N, 0.f); st.n_valid_contrib.assign(N, 0); }
    std::vector<double> unc_acc;
    if (LIGHT && gau_unc) unc_acc.assign(st.P, 0.0);
    const float* depths = st.depths.data();
    const int tiles = st.gx * st.gy;
#pragma omp parallel for schedule(dynamic, 4)
    for (int tile = 0; tile < tiles; tile++) {
        const uint32_t r0 = st.ranges[2 * tile], r1 = st.ranges[2 * tile + 1];
        const int tx = tile % st.gx, ty = tile / st.gx;
        for (int ly = 0; ly < BLOCK_Y; ly++)
            for (int lx = 0; lx < BLOCK_X; lx++) {
                const uint32_t px = tx * BLOCK_X + lx, py = ty * BLOCK_Y + ly;
                if (!(px < (uint32_t)W && py < (uint32_t)H)) continue;
                const size_t pix_id = (size_t)W * py + px;
                const float pfx = (float)px, pfy = (float)py;
                float T = 1.0f;
                uint32_t contributor = 0, last_contributor = 0, valid_contributor = 0;
                float C[3] = {0, 0, 0};
                float weight = 0, Dd = 0, D_median = 0.0f, D_var = 0.0f, U = 0;
                const float gt_px = (LIGHT && gt_depth) ? gt_depth[pix_id] : 0.f;
                for (uint32_t k = r0; k < r1; k++) {
                    contributor++;
                    const uint32_t id = st.point_list[k];
                    const float xyx = st.means2D[2 * (size_t)id], xyy = st.means2D[2 * (size_t)id + 1];
                    const float dx = xyx - pfx, dy = xyy - pfy;
                    const float* co = &st.conic_opacity[4 * (size_t)id];
                    const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                    if (power > 0.0f) continue;
                    const float alpha = std::min<decltype(co[3] * m_exp(power))>(0.99f, co[3] * m_exp(power));
                    if (alpha < 15.0f / 255.0f) continue;
                    if (LIGHT) {
                        const float test_T = T * (1 - alpha);
                        if (test_T < 0.0001f) break;  // done = true (this Gaussian is NOT blended)
                        for (int ch = 0; ch < 3; ch++) C[ch] += features[(size_t)id * 3 + ch] * alpha * T;
                        weight += alpha * T;
                        Dd += depths[id] * alpha * T;
                        if (T > 0.5f && test_T < 0.5) {
                            D_median = depths[id];
                            if (gau_unc) {
                                const float u = ((depths[id] - gt_px)) * (depths[id] - gt_px) * alpha * T;
#pragma omp atomic
                                unc_acc[id] += (double)u;
                            }
                            if (gau_px) {
#pragma omp atomic
                                gau_px[id] += 1;
                            }
                        }
                        T = test_T;
                        last_contributor = contributor;
                    } else {
                        for (int ch = 0; ch < 3; ch++) C[ch] += features[(size_t)id * 3 + ch] * alpha * T;
                        Dd += depths[id] * alpha * T;
                        U += alpha * T;
                        valid_contributor++;
                        T = T * (1 - alpha);
                        last_contributor = contributor;
                        if (T < 0.0001f) break;  // blended first, then done (F/cr/forward.cu:370-381)
                    }
                }
                st.n_contrib[pix_id] = last_contributor;
                for (int ch = 0; ch < 3; ch++) out_color[ch * N + pix_id] = C[ch] + T * bg[ch];
                out_depth[pix_id] = Dd;
                if (LIGHT) {
                    out_alpha_or_unc[pix_id] = weight;
                    out_median[pix_id] = D_median;
                    out_depth_var[pix_id] = D_var;
                } else {
                    st.final_T[pix_id] = T;
                    st.n_valid_contrib[pix_id] = valid_contributor;
                    out_alpha_or_unc[pix_id] = U;
                }
            }
    }
    if (LIGHT && gau_unc)
        for (int i = 0; i < st.P; i++) gau_unc[i] = (float)unc_acc[i];
}

// ---------------------------------------------------------------- per-Gaussian backward
// L/cr/backward.cu:20-139
void shBackward(int idx, int deg, int M, const float* means, V3 campos, const float* shs, const uint8_t* clamped,
                const float* dL_dcolor, float* dL_dmeans, float* dL_dshs, float* dgc_dCampos = nullptr) {
    V3 pos = {means[3 * idx], means[3 * idx + 1], means[3 * idx + 2]};
    V3 dir_orig = pos - campos;
    V3 dir = dir_orig / length(dir_orig);
    // full variant only: d(dir)/d(campos) (F/cr/backward.cu:27-43)
    float dxdCamx = 0, dydCamx = 0, dzdCamx = 0, dxdCamy = 0, dydCamy = 0, dzdCamy = 0, dxdCamz = 0, dydCamz = 0, dzdCamz = 0;
    if (dgc_dCampos) {
        float len_dir_orig3 = length(dir_orig) * length(dir_orig) * length(dir_orig);
        float one_len_dir_orig3 = 1.0f / len_dir_orig3;
        float one_len_dir_orig = 1.0f / length(dir_orig);
        dxdCamx = dir_orig.x * dir_orig.x * one_len_dir_orig3 - one_len_dir_orig;
        dydCamx = dir_orig.x * dir_orig.y * one_len_dir_orig3;
        dzdCamx = dir_orig.x * dir_orig.z * one_len_dir_orig3;
        dxdCamy = dir_orig.x * dir_orig.y * one_len_dir_orig3;
        dydCamy = dir_orig.y * dir_orig.y * one_len_dir_orig3 - one_len_dir_orig;
        dzdCamy = dir_orig.y * dir_orig.z * one_len_dir_orig3;
        dxdCamz = dir_orig.x * dir_orig.z * one_len_dir_orig3;
        dydCamz = dir_orig.y * dir_orig.z * one_len_dir_orig3;
        dzdCamz = dir_orig.z * dir_orig.z * one_len_dir_orig3 - one_len_dir_orig;
    }
    const V3* sh = reinterpret_cast<const V3*>(shs) + (size_t)idx * M;
    V3 dL_dRGB = {dL_dcolor[3 * idx], dL_dcolor[3 * idx + 1], dL_dcolor[3 * idx + 2]};
    dL_dRGB.x *= clamped[3 * idx + 0] ? 0 : 1;
    dL_dRGB.y *= clamped[3 * idx + 1] ? 0 : 1;
    dL_dRGB.z *= clamped[3 * idx + 2] ? 0 : 1;
    V3 dRGBdx = {0, 0, 0}, dRGBdy = {0, 0, 0}, dRGBdz = {0, 0, 0};
    float x = dir.x, y = dir.y, z = dir.z;
    V3* dL_dsh = reinterpret_cast<V3*>(dL_dshs) + (size_t)idx * M;
    float dRGBdsh0 = SH_C0;
    dL_dsh[0] = dRGBdsh0 * dL_dRGB;
    if (deg > 0) {
        float dRGBdsh1 = -SH_C1 * y, dRGBdsh2 = SH_C1 * z, dRGBdsh3 = -SH_C1 * x;
        dL_dsh[1] = dRGBdsh1 * dL_dRGB;
        dL_dsh[2] = dRGBdsh2 * dL_dRGB;
        dL_dsh[3] = dRGBdsh3 * dL_dRGB;
        dRGBdx = -SH_C1 * sh[3];
        dRGBdy = -SH_C1 * sh[1];
        dRGBdz = SH_C1 * sh[2];
        if (deg > 1) {
            float xx = x * x, yy = y * y, zz = z * z;
            float xy = x * y, yz = y * z, xz = x * z;
            float d4 = SH_C2[0] * xy, d5 = SH_C2[1] * yz, d6 = SH_C2[2] * (2.f * zz - xx - yy);
            float d7 = SH_C2[3] * xz, d8 = SH_C2[4] * (xx - yy);
            dL_dsh[4] = d4 * dL_dRGB; dL_dsh[5] = d5 * dL_dRGB; dL_dsh[6] = d6 * dL_dRGB;
            dL_dsh[7] = d7 * dL_dRGB; dL_dsh[8] = d8 * dL_dRGB;
            dRGBdx += SH_C2[0] * y * sh[4] + SH_C2[2] * 2.f * -x * sh[6] + SH_C2[3] * z * sh[7] + SH_C2[4] * 2.f * x * sh[8];
            dRGBdy += SH_C2[0] * x * sh[4] + SH_C2[1] * z * sh[5] + SH_C2[2] * 2.f * -y * sh[6] + SH_C2[4] * 2.f * -y * sh[8];
            dRGBdz += SH_C2[1] * y * sh[5] + SH_C2[2] * 2.f * 2.f * z * sh[6] + SH_C2[3] * x * sh[7];
            if (deg > 2) {
                float d9 = SH_C3[0] * y * (3.f * xx - yy), d10 = SH_C3[1] * xy * z;
                float d11 = SH_C3[2] * y * (4.f * zz - xx - yy);
                float d12 = SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy);
                float d13 = SH_C3[4] * x * (4.f * zz - xx - yy), d14 = SH_C3[5] * z * (xx - yy);
                float d15 = SH_C3[6] * x * (xx - 3.f * yy);
                dL_dsh[9] = d9 * dL_dRGB; dL_dsh[10] = d10 * dL_dRGB; dL_dsh[11] = d11 * dL_dRGB;
                dL_dsh[12] = d12 * dL_dRGB; dL_dsh[13] = d13 * dL_dRGB; dL_dsh[14] = d14 * dL_dRGB;
                dL_dsh[15] = d15 * dL_dRGB;
                dRGBdx += (SH_C3[0] * sh[9] * 3.f * 2.f * xy + SH_C3[1] * sh[10] * yz + SH_C3[2] * sh[11] * -2.f * xy +
                           SH_C3[3] * sh[12] * -3.f * 2.f * xz + SH_C3[4] * sh[13] * (-3.f * xx + 4.f * zz - yy) +
                           SH_C3[5] * sh[14] * 2.f * xz + SH_C3[6] * sh[15] * 3.f * (xx - yy));
                dRGBdy += (SH_C3[0] * sh[9] * 3.f * (xx - yy) + SH_C3[1] * sh[10] * xz +
                           SH_C3[2] * sh[11] * (-3.f * yy + 4.f * zz - xx) + SH_C3[3] * sh[12] * -3.f * 2.f * yz +
                           SH_C3[4] * sh[13] * -2.f * xy + SH_C3[5] * sh[14] * -2.f * yz +
                           SH_C3[6] * sh[15] * -3.f * 2.f * xy);
                dRGBdz += (SH_C3[1] * sh[10] * xy + SH_C3[2] * sh[11] * 4.f * 2.f * yz +
                           SH_C3[3] * sh[12] * 3.f * (2.f * zz - xx - yy) + SH_C3[4] * sh[13] * 4.f * 2.f * xz +
                           SH_C3[5] * sh[14] * (xx - yy));
            }
        }
    }
    if (dgc_dCampos) {  // F/cr/backward.cu:159-166 (not clamp-masked)
        V3* o = reinterpret_cast<V3*>(dgc_dCampos) + (size_t)idx * 3;
        o[0] = dRGBdx * dxdCamx + dRGBdy * dydCamx + dRGBdz * dzdCamx;
        o[1] = dRGBdx * dxdCamy + dRGBdy * dydCamy + dRGBdz * dzdCamy;
        o[2] = dRGBdx * dxdCamz + dRGBdy * dydCamz + dRGBdz * dzdCamz;
    }
    V3 dL_ddir = {dot(dRGBdx, dL_dRGB), dot(dRGBdy, dL_dRGB), dot(dRGBdz, dL_dRGB)};
    V3 dL_dmean = dnormvdv(dir_orig, dL_ddir);
    dL_dmeans[3 * idx + 0] += dL_dmean.x;
    dL_dmeans[3 * idx + 1] += dL_dmean.y;
    dL_dmeans[3 * idx + 2] += dL_dmean.z;
}

// L/cr/backward.cu:144-276 (light: accumulates into dL_dmeans)
// full: dL_dgau_depths != nullptr -> dL_dmeans is ASSIGNED and the depth term added (F/cr/backward.cu:383-386);
// its dginvcovs_dT output (F:243-312) feeds only ComputePG's part 2-2, which is never summed (dead) and is skipped.
void cov2DBackwardLight(const State& st, int P, const float* means, const int32_t* radii, const float* cov3Ds,
                        float hx, float hy, float tanx, float tany, const float* view, const float* dL_dconics,
                        float* dL_dmeans, float* dL_dcov, const float* dL_dgau_depths = nullptr) {
#pragma omp parallel for schedule(static)
    for (int idx = 0; idx < P; idx++) {
        if (!(radii[idx] > 0)) continue;
        const float* c3 = cov3Ds + 6 * (size_t)idx;
        V3 mean = {means[3 * idx], means[3 * idx + 1], means[3 * idx + 2]};
        V3 dL_dconic = {dL_dconics[4 * (size_t)idx], dL_dconics[4 * (size_t)idx + 1], dL_dconics[4 * (size_t)idx + 3]};
        Cov2DFwd c;
        cov2DCommon(mean, hx, hy, tanx, tany, c3, view, c);
        const float limx = 1.3f * tanx, limy = 1.3f * tany;
        const float x_grad_mul = c.txtz < -limx || c.txtz > limx ? 0 : 1;
        const float y_grad_mul = c.tytz < -limy || c.tytz > limy ? 0 : 1;
        const M3& T = c.T; const M3& Vrk = c.Vrk; const M3& Wm = c.Wm; const V3 t = c.t;
        float a = c.cov[0][0] += 0.3f;
        float b = c.cov[0][1];
        float cc = c.cov[1][1] += 0.3f;
        float denom = a * cc - b * b;
        float dL_da = 0, dL_db = 0, dL_dc = 0;
        float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
        float* dc = dL_dcov + 6 * (size_t)idx;
        if (denom2inv != 0) {
            dL_da = denom2inv * (-cc * cc * dL_dconic.x + 2 * b * cc * dL_dconic.y + (denom - a * cc) * dL_dconic.z);
            dL_dc = denom2inv * (-a * a * dL_dconic.z + 2 * a * b * dL_dconic.y + (denom - a * cc) * dL_dconic.x);
            dL_db = denom2inv * 2 * (b * cc * dL_dconic.x - (denom + 2 * b * b) * dL_dconic.y + a * b * dL_dconic.z);
            dc[0] = (T[0][0] * T[0][0] * dL_da + T[0][0] * T[1][0] * dL_db + T[1][0] * T[1][0] * dL_dc);
            dc[3] = (T[0][1] * T[0][1] * dL_da + T[0][1] * T[1][1] * dL_db + T[1][1] * T[1][1] * dL_dc);
            dc[5] = (T[0][2] * T[0][2] * dL_da + T[0][2] * T[1][2] * dL_db + T[1][2] * T[1][2] * dL_dc);
            dc[1] = 2 * T[0][0] * T[0][1] * dL_da + (T[0][0] * T[1][1] + T[0][1] * T[1][0]) * dL_db + 2 * T[1][0] * T[1][1] * dL_dc;
            dc[2] = 2 * T[0][0] * T[0][2] * dL_da + (T[0][0] * T[1][2] + T[0][2] * T[1][0]) * dL_db + 2 * T[1][0] * T[1][2] * dL_dc;
            dc[4] = 2 * T[0][2] * T[0][1] * dL_da + (T[0][1] * T[1][2] + T[0][2] * T[1][1]) * dL_db + 2 * T[1][1] * T[1][2] * dL_dc;
        } else {
            for (int i = 0; i < 6; i++) dc[i] = 0;
        }
        float dL_dT00 = 2 * (T[0][0] * Vrk[0][0] + T[0][1] * Vrk[0][1] + T[0][2] * Vrk[0][2]) * dL_da +
                        (T[1][0] * Vrk[0][0] + T[1][1] * Vrk[0][1] + T[1][2] * Vrk[0][2]) * dL_db;
        float dL_dT01 = 2 * (T[0][0] * Vrk[1][0] + T[0][1] * Vrk[1][1] + T[0][2] * Vrk[1][2]) * dL_da +
                        (T[1][0] * Vrk[1][0] + T[1][1] * Vrk[1][1] + T[1][2] * Vrk[1][2]) * dL_db;
        float dL_dT02 = 2 * (T[0][0] * Vrk[2][0] + T[0][1] * Vrk[2][1] + T[0][2] * Vrk[2][2]) * dL_da +
                        (T[1][0] * Vrk[2][0] + T[1][1] * Vrk[2][1] + T[1][2] * Vrk[2][2]) * dL_db;
        float dL_dT10 = 2 * (T[1][0] * Vrk[0][0] + T[1][1] * Vrk[0][1] + T[1][2] * Vrk[0][2]) * dL_dc +
                        (T[0][0] * Vrk[0][0] + T[0][1] * Vrk[0][1] + T[0][2] * Vrk[0][2]) * dL_db;
        float dL_dT11 = 2 * (T[1][0] * Vrk[1][0] + T[1][1] * Vrk[1][1] + T[1][2] * Vrk[1][2]) * dL_dc +
                        (T[0][0] * Vrk[1][0] + T[0][1] * Vrk[1][1] + T[0][2] * Vrk[1][2]) * dL_db;
        float dL_dT12 = 2 * (T[1][0] * Vrk[2][0] + T[1][1] * Vrk[2][1] + T[1][2] * Vrk[2][2]) * dL_dc +
                        (T[0][0] * Vrk[2][0] + T[0][1] * Vrk[2][1] + T[0][2] * Vrk[2][2]) * dL_db;
        float dL_dJ00 = Wm[0][0] * dL_dT00 + Wm[0][1] * dL_dT01 + Wm[0][2] * dL_dT02;
        float dL_dJ02 = Wm[2][0] * dL_dT00 + Wm[2][1] * dL_dT01 + Wm[2][2] * dL_dT02;
        float dL_dJ11 = Wm[1][0] * dL_dT10 + Wm[1][1] * dL_dT11 + Wm[1][2] * dL_dT12;
        float dL_dJ12 = Wm[2][0] * dL_dT10 + Wm[2][1] * dL_dT11 + Wm[2][2] * dL_dT12;
        float tz = 1.f / t.z;
        float tz2 = tz * tz;
        float tz3 = tz2 * tz;
        float dL_dtx = x_grad_mul * -hx * tz2 * dL_dJ02;
        float dL_dty = y_grad_mul * -hy * tz2 * dL_dJ12;
        float dL_dtz = -hx * tz2 * dL_dJ00 - hy * tz2 * dL_dJ11 + (2 * hx * t.x) * tz3 * dL_dJ02 + (2 * hy * t.y) * tz3 * dL_dJ12;
        V3 dL_dmean = transformVec4x3Transpose({dL_dtx, dL_dty, dL_dtz}, view);
        if (dL_dgau_depths) {
            const float dgd = dL_dgau_depths[idx];
            float mul3 = view[2] * mean.x + view[6] * mean.y + view[10] * mean.z + view[14];
            dL_dmeans[3 * idx + 0] = dL_dmean.x + dgd * (view[2] - view[3] * mul3);
            dL_dmeans[3 * idx + 1] = dL_dmean.y + dgd * (view[6] - view[7] * mul3);
            dL_dmeans[3 * idx + 2] = dL_dmean.z + dgd * (view[10] - view[11] * mul3);
        } else {
            dL_dmeans[3 * idx + 0] += dL_dmean.x;
            dL_dmeans[3 * idx + 1] += dL_dmean.y;
            dL_dmeans[3 * idx + 2] += dL_dmean.z;
        }
    }
    (void)st;
}

// L/cr/backward.cu:280-343
void cov3DBackward(int idx, V3 scale, float mod, const float* rot, const float* dL_dcov3Ds, float* dL_dscales,
                   float* dL_drots) {
    float r = rot[0], x = rot[1], y = rot[2], z = rot[3];
    M3 R = mat3(1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
                2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
                2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y));
    M3 S = mat3(1, 0, 0, 0, 1, 0, 0, 0, 1);
    V3 s = mod * scale;
    S[0][0] = s.x; S[1][1] = s.y; S[2][2] = s.z;
    M3 Mm = S * R;
    const float* d = dL_dcov3Ds + 6 * (size_t)idx;
    M3 dL_dSigma = mat3(d[0], 0.5f * d[1], 0.5f * d[2], 0.5f * d[1], d[3], 0.5f * d[4], 0.5f * d[2], 0.5f * d[4], d[5]);
    M3 dL_dM = 2.0f * Mm * dL_dSigma;
    M3 Rt = transpose(R);
    M3 dL_dMt = transpose(dL_dM);
    float* ds = dL_dscales + 3 * (size_t)idx;
    ds[0] = dot(col(Rt, 0), col(dL_dMt, 0));
    ds[1] = dot(col(Rt, 1), col(dL_dMt, 1));
    ds[2] = dot(col(Rt, 2), col(dL_dMt, 2));
    for (int k = 0; k < 3; k++) { dL_dMt[0][k] *= s.x; dL_dMt[1][k] *= s.y; dL_dMt[2][k] *= s.z; }
    float* dq = dL_drots + 4 * (size_t)idx;
    dq[0] = 2 * z * (dL_dMt[0][1] - dL_dMt[1][0]) + 2 * y * (dL_dMt[2][0] - dL_dMt[0][2]) + 2 * x * (dL_dMt[1][2] - dL_dMt[2][1]);
    dq[1] = 2 * y * (dL_dMt[1][0] + dL_dMt[0][1]) + 2 * z * (dL_dMt[2][0] + dL_dMt[0][2]) + 2 * r * (dL_dMt[1][2] - dL_dMt[2][1]) - 4 * x * (dL_dMt[2][2] + dL_dMt[1][1]);
    dq[2] = 2 * x * (dL_dMt[1][0] + dL_dMt[0][1]) + 2 * r * (dL_dMt[2][0] - dL_dMt[0][2]) + 2 * z * (dL_dMt[1][2] + dL_dMt[2][1]) - 4 * y * (dL_dMt[2][2] + dL_dMt[0][0]);
    dq[3] = 2 * r * (dL_dMt[0][1] - dL_dMt[1][0]) + 2 * x * (dL_dMt[2][0] + dL_dMt[0][2]) + 2 * y * (dL_dMt[1][2] + dL_dMt[2][1]) - 4 * z * (dL_dMt[1][1] + dL_dMt[0][0]);
}

// L/cr/backward.cu:348-416
void preprocessBackwardLight(int P, int D, int M, const float* means, const int32_t* radii, const float* shs,
                             const uint8_t* clamped, const float* scales, const float* rots, float mod,
                             const float* view, const float* proj, const float* campos, const float* dL_dmean2D,
                             float* dL_dmeans, const float* dL_dcolor, const float* dL_ddepth, const float* dL_dcov3D,
                             float* dL_dsh, float* dL_dscale, float* dL_drot, float* dgc_dCampos = nullptr) {
    // dL_ddepth == nullptr selects the full flavour (F/cr/backward.cu:459-537): no depth->mean term here
    V3 cam = {0, 0, 0};
    if (campos) cam = {campos[0], campos[1], campos[2]};
#pragma omp parallel for schedule(static)
    for (int idx = 0; idx < P; idx++) {
        if (!(radii[idx] > 0)) continue;
        V3 m = {means[3 * idx], means[3 * idx + 1], means[3 * idx + 2]};
        V4 m_hom = transformPoint4x4(m, proj);
        float m_w = 1.0f / (m_hom.w + 0.0000001f);
        const float g2x = dL_dmean2D[3 * (size_t)idx], g2y = dL_dmean2D[3 * (size_t)idx + 1];
        float mul1 = (proj[0] * m.x + proj[4] * m.y + proj[8] * m.z + proj[12]) * m_w * m_w;
        float mul2 = (proj[1] * m.x + proj[5] * m.y + proj[9] * m.z + proj[13]) * m_w * m_w;
        V3 dL_dmean;
        dL_dmean.x = (proj[0] * m_w - proj[3] * mul1) * g2x + (proj[1] * m_w - proj[3] * mul2) * g2y;
        dL_dmean.y = (proj[4] * m_w - proj[7] * mul1) * g2x + (proj[5] * m_w - proj[7] * mul2) * g2y;
        dL_dmean.z = (proj[8] * m_w - proj[11] * mul1) * g2x + (proj[9] * m_w - proj[11] * mul2) * g2y;
        dL_dmeans[3 * idx + 0] += dL_dmean.x;
        dL_dmeans[3 * idx + 1] += dL_dmean.y;
        dL_dmeans[3 * idx + 2] += dL_dmean.z;
        if (dL_ddepth) {
            float mul3 = view[2] * m.x + view[6] * m.y + view[10] * m.z + view[14];
            V3 d2;
            d2.x = (view[2] - view[3] * mul3) * dL_ddepth[idx];
            d2.y = (view[6] - view[7] * mul3) * dL_ddepth[idx];
            d2.z = (view[10] - view[11] * mul3) * dL_ddepth[idx];
            dL_dmeans[3 * idx + 0] += d2.x;
            dL_dmeans[3 * idx + 1] += d2.y;
            dL_dmeans[3 * idx + 2] += d2.z;
        }
        if (shs) shBackward(idx, D, M, means, cam, shs, clamped, dL_dcolor, dL_dmeans, dL_dsh, dgc_dCampos);
        if (scales)
            cov3DBackward(idx, {scales[3 * idx], scales[3 * idx + 1], scales[3 * idx + 2]}, mod, rots + 4 * (size_t)idx,
                          dL_dcov3D, dL_dscale, dL_drot);
    }
}

// L/cr/backward.cu:701-751
void poseGradientPre(State& st, int P, const float* means, const int32_t* radii, const float* proj,
                     const float* perspec) {
    st.dgndcs_dview.assign(24 * (size_t)P, 0.f);
    st.dg_camd.assign(4 * (size_t)P, 0.f);
#pragma omp parallel for schedule(static)
    for (int idx = 0; idx < P; idx++) {
        if (!(radii[idx] > 0)) continue;
        V3 m = {means[3 * idx], means[3 * idx + 1], means[3 * idx + 2]};
        V4 m_hom = transformPoint4x4(m, proj);
        float m_w = 1.0f / (m_hom.w + 0.0000001f);
        float* g = &st.dgndcs_dview[24 * (size_t)idx];  // [12][2]
        const float mm[4] = {m.x, m.y, m.z, 1.0f};
        for (int k = 0; k < 4; k++) {
            g[(3 * k + 0) * 2 + 0] = m_w * perspec[0] * mm[k];
            g[(3 * k + 2) * 2 + 0] = m_hom.x * (-m_w * m_w) * mm[k];
            g[(3 * k + 1) * 2 + 1] = m_w * perspec[5] * mm[k];
            g[(3 * k + 2) * 2 + 1] = m_hom.y * (-m_w * m_w) * mm[k];
            st.dg_camd[4 * (size_t)idx + k] = mm[k];
        }
    }
}

// ---------------------------------------------------------------- backward blend, light
// L/cr/backward.cu:419-699.  acc[g*13 + {0..2 colour, 3 depth, 4..5 mean2D, 6..8 conic xyw,
// 9 opacity, 10..12 median->mean3D}] are double accumulators standing in for atomicAdd.
void renderBackwardLight(const State& st, const float* bg, const float* colors, const float* alphas,
                         const float* dL_dpixels, const float* dL_dpix_depth, const float* dL_dpix_median,
                         const float* dL_dpix_var, const float* means, const float* view, const float* gt_depth,
                         bool track_off, bool map_off, std::vector<double>& acc, float* dL_dview_pix,
                         double* dL_dview_sum) {
    const int W = st.W, H = st.H;
    const size_t N = (size_t)W * H;
    const int tiles = st.gx * st.gy;
    const float ddelx_dx = 0.5 * W, ddely_dy = 0.5 * H;
    const float* depths = st.depths.data();
    double vsum[16] = {0};
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : vsum[:16])
    for (int tile = 0; tile < tiles; tile++) {
        const uint32_t r0 = st.ranges[2 * tile], r1 = st.ranges[2 * tile + 1];
        const int tx = tile % st.gx, ty = tile / st.gx;
        for (int ly = 0; ly < BLOCK_Y; ly++)
            for (int lx = 0; lx < BLOCK_X; lx++) {
                const uint32_t px = tx * BLOCK_X + lx, py = ty * BLOCK_Y + ly;
                if (!(px < (uint32_t)W && py < (uint32_t)H)) continue;
                const size_t pix_id = (size_t)W * py + px;
                const float pfx = (float)px, pfy = (float)py;
                const float T_final = 1 - alphas[pix_id];
                float T = T_final;
                uint32_t contributor = r1 - r0;
                const int last_contributor = st.n_contrib[pix_id];
                float accum_rec[3] = {0, 0, 0}, dL_dpixel[3];
                float accum_depth_rec = 0, accum_var_rec = 0;
                for (int i = 0; i < 3; i++) dL_dpixel[i] = dL_dpixels[i * N + pix_id];
                const float dL_dpixel_depth = dL_dpix_depth[pix_id];
                const float dL_dpixel_median_depth = dL_dpix_median[pix_id];
                const float dL_dpixel_depth_var = dL_dpix_var[pix_id];
                const float gt_px_depth = gt_depth[pix_id];
                float last_alpha = 0, last_color[3] = {0, 0, 0}, last_depth = 0, last_var = 0;
                float dv[12] = {0};
                bool mid_once = true;
                for (uint32_t k = r1; k-- > r0;) {
                    contributor--;
                    if ((int)contributor >= last_contributor) continue;
                    const uint32_t gid = st.point_list[k];
                    const float xyx = st.means2D[2 * (size_t)gid], xyy = st.means2D[2 * (size_t)gid + 1];
                    const float dx = xyx - pfx, dy = xyy - pfy;
                    const float* co = &st.conic_opacity[4 * (size_t)gid];
                    const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                    if (power > 0.0f) continue;
                    const float G = m_exp(power);
                    const float alpha = std::min(0.99f, co[3] * G);
                    if (alpha < 15.0f / 255.0f) continue;
                    T = T / (1.f - alpha);
                    const float dchannel_dcolor = alpha * T;
                    const float dpixel_depth_ddepth = alpha * T;
                    const float dL_ddepth = dpixel_depth_ddepth * dL_dpixel_depth;
                    float dL_dalpha = 0.0f;
                    double* a = &acc[(size_t)gid * 13];
                    for (int ch = 0; ch < 3; ch++) {
                        const float c = colors[(size_t)gid * 3 + ch];
                        accum_rec[ch] = last_alpha * last_color[ch] + (1.f - last_alpha) * accum_rec[ch];
                        last_color[ch] = c;
                        const float dL_dchannel = dL_dpixel[ch];
                        dL_dalpha += (c - accum_rec[ch]) * dL_dchannel;
                        if (!map_off) {
                            const float v = dchannel_dcolor * dL_dchannel;
#pragma omp atomic
                            a[ch] += (double)v;
                        }
                    }
                    const float c_d = depths[gid];
                    const float c_var = (c_d - gt_px_depth) * (c_d - gt_px_depth);
                    accum_depth_rec = last_alpha * last_depth + (1.f - last_alpha) * accum_depth_rec;
                    last_depth = c_d;
                    accum_var_rec = last_alpha * last_var + (1.f - last_alpha) * accum_var_rec;
                    last_var = c_var;
                    dL_dalpha += (c_d - accum_depth_rec) * dL_dpixel_depth;
                    dL_dalpha += (c_var - accum_var_rec) * dL_dpixel_depth_var;
                    if (!map_off) {
                        const float v = dL_ddepth + dL_dpixel_depth_var * dpixel_depth_ddepth * 2. * (c_d - gt_px_depth);
#pragma omp atomic
                        a[3] += (double)v;
                    }
                    dL_dalpha *= T;
                    last_alpha = alpha;
                    float bg_dot_dpixel = 0;
                    for (int i = 0; i < 3; i++) bg_dot_dpixel += bg[i] * dL_dpixel[i];
                    dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot_dpixel;
                    const float dL_dG = co[3] * dL_dalpha;
                    const float gdx = G * dx, gdy = G * dy;
                    const float dG_ddelx = -gdx * co[0] - gdy * co[1];
                    const float dG_ddely = -gdy * co[2] - gdx * co[1];
                    if (!track_off) {
                        const float nx = dL_dalpha * co[3] * dG_ddelx * ddelx_dx;
                        const float ny = dL_dalpha * co[3] * dG_ddely * ddely_dy;
                        const float* J = &st.dgndcs_dview[24 * (size_t)gid];
                        const float* cd = &st.dg_camd[4 * (size_t)gid];
                        for (int s = 0; s < 12; s++) {
                            if (s % 3 == 2) dv[s] += J[2 * s] * nx + J[2 * s + 1] * ny + cd[s / 3] * dL_ddepth;
                            else dv[s] += J[2 * s] * nx + J[2 * s + 1] * ny;
                        }
                    }
                    if (!map_off) {
                        if (T > 0.5f && mid_once) {
                            const float* mg = means + 3 * (size_t)gid;
                            float mul3 = view[2] * mg[0] + view[6] * mg[1] + view[10] * mg[2] + view[14];
                            const float vx = (view[2] - view[3] * mul3) * dL_dpixel_median_depth * 1.0f;
                            const float vy = (view[6] - view[7] * mul3) * dL_dpixel_median_depth * 1.0f;
                            const float vz = (view[10] - view[11] * mul3) * dL_dpixel_median_depth * 1.0f;
#pragma omp atomic
                            a[10] += (double)vx;
#pragma omp atomic
                            a[11] += (double)vy;
#pragma omp atomic
                            a[12] += (double)vz;
                            mid_once = false;
                        }
                        const float v4 = dL_dG * dG_ddelx * ddelx_dx, v5 = dL_dG * dG_ddely * ddely_dy;
                        const float v6 = -0.5f * gdx * dx * dL_dG, v7 = -0.5f * gdx * dy * dL_dG;
                        const float v8 = -0.5f * gdy * dy * dL_dG, v9 = G * dL_dalpha;
#pragma omp atomic
                        a[4] += (double)v4;
#pragma omp atomic
                        a[5] += (double)v5;
#pragma omp atomic
                        a[6] += (double)v6;
#pragma omp atomic
                        a[7] += (double)v7;
#pragma omp atomic
                        a[8] += (double)v8;
#pragma omp atomic
                        a[9] += (double)v9;
                    }
                }
                if (!track_off) {
                    static const int slot2entry[12] = {0, 1, 2, 4, 5, 6, 8, 9, 10, 12, 13, 14};
                    for (int s = 0; s < 12; s++) {
                        if (dL_dview_pix) dL_dview_pix[pix_id * 16 + slot2entry[s]] = dv[s];
                        vsum[slot2entry[s]] += (double)dv[s];
                    }
                }
            }
    }
    for (int i = 0; i < 16; i++) dL_dview_sum[i] = vsum[i];
}


// ---------------------------------------------------------------- backward blend + pose gradient, full
// F/cr/backward.cu:540-836 (renderCUDA) and :838-1338 (ComputePG).  The reference writes 23 words per valid
// (pixel, Gaussian) pair into NG-sized lists and re-walks every tile to consume them; here each pixel keeps
// its own pairs in a local vector and ComputePG's per-pixel body runs right after the blend loop -- same
// values, same order (back to front), same arithmetic:
//   * only part 1 (colour -> campos -> view) and part 2-1 (ndc -> view) are summed (:1264-1275); part 2-2
//     (conic -> T -> view, :1076-1243) is computed there but never used, so it is not restated;
//   * the depth terms dd_dvK are ASSIGNED, not accumulated (:1278-1289): only the last matched pair --
//     the front-most valid Gaussian of the pixel -- contributes dL_depth * dd_dvK;
//   * dL_duncertainty never enters the pose gradient; entries 3,7,11,15 are never written.
// acc[g*10 + {0..2 colour, 3 gau depth, 4..5 mean2D, 6..8 conic, 9 opacity}]
int g_stale_collected_id[256];  // emulate_dropout only: ComputePG's __shared__ collected_id across blocks
struct PairRec {
    uint32_t gid;
    float dpix_dgc;
    float dndcs[2][3];
    float ddepth_dndcs[2];
};
void renderBackwardFull(const State& st, const float* bg, const float* colors, const float* dL_dpixels,
                        const float* dL_depths, const float* dL_duncertainties, const float* gt_depth,
                        std::vector<double>& acc, const float* means, const float* view, const float* dgc_dCampos_or_null,
                        bool pose_pass, double* dL_dview_sum, bool emulate_dropout = false) {
    const int W = st.W, H = st.H;
    const size_t N = (size_t)W * H;
    const int tiles = st.gx * st.gy;
    const float ddelx_dx = 0.5 * W, ddely_dy = 0.5 * H;
    const float* depths = st.depths.data();
    double vsum[16] = {0};
    if (emulate_dropout) std::fill(g_stale_collected_id, g_stale_collected_id + 256, 0);
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : vsum[:16]) if (!emulate_dropout)
    for (int tile = 0; tile < tiles; tile++) {
        const uint32_t r0 = st.ranges[2 * tile], r1 = st.ranges[2 * tile + 1];
        const int tx = tile % st.gx, ty = tile / st.gx;
        std::vector<PairRec> pairs;
        std::vector<std::vector<PairRec>> tile_pairs(pose_pass ? 256 : 0);
        for (int ly = 0; ly < BLOCK_Y; ly++)
            for (int lx = 0; lx < BLOCK_X; lx++) {
                const uint32_t px = tx * BLOCK_X + lx, py = ty * BLOCK_Y + ly;
                if (!(px < (uint32_t)W && py < (uint32_t)H)) continue;
                const size_t pix_id = (size_t)W * py + px;
                const float pfx = (float)px, pfy = (float)py;
                const float T_final = st.final_T[pix_id];
                float T = T_final;
                uint32_t contributor = r1 - r0;
                const int last_contributor = st.n_contrib[pix_id];
                float dpixel_dalpha[3] = {0, 0, 0}, ddepth_dalpha = 0;
                float accum_rec[3] = {0, 0, 0}, dL_dpixel[3];
                for (int i = 0; i < 3; i++) dL_dpixel[i] = dL_dpixels[i * N + pix_id];
                const float dL_depth = dL_depths[pix_id];
                const float dL_duncertainty = dL_duncertainties[pix_id];
                const float gt_px_depth = gt_depth[pix_id];
                float accum_depth_rec = 0, accum_uncertainty_rec = 0;
                float last_alpha = 0, last_color[3] = {0, 0, 0}, last_depth = 0, last_uncertainty = 0;
                pairs.clear();
                for (uint32_t k = r1; k-- > r0;) {
                    contributor--;
                    if ((int)contributor >= last_contributor) continue;
                    const uint32_t gid = st.point_list[k];
                    const float dx = st.means2D[2 * (size_t)gid] - pfx, dy = st.means2D[2 * (size_t)gid + 1] - pfy;
                    const float* co = &st.conic_opacity[4 * (size_t)gid];
                    const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                    if (power > 0.0f) continue;
                    const float G = m_exp(power);
                    const float alpha = std::min(0.99f, co[3] * G);
                    if (alpha < 15.0f / 255.0f) continue;
                    T = T / (1.f - alpha);
                    const float dchannel_dcolor = alpha * T;
                    float dL_dalpha = 0.0f;
                    double* a = pose_pass ? nullptr : &acc[(size_t)gid * 10];
                    for (int ch = 0; ch < 3; ch++) {
                        const float c = colors[(size_t)gid * 3 + ch];
                        accum_rec[ch] = last_alpha * last_color[ch] + (1.f - last_alpha) * accum_rec[ch];
                        last_color[ch] = c;
                        const float dL_dchannel = dL_dpixel[ch];
                        dL_dalpha += (c - accum_rec[ch]) * dL_dchannel;
                        dpixel_dalpha[ch] = T * (c - accum_rec[ch]);
                        if (a) {
                            const float v = dchannel_dcolor * dL_dchannel;
#pragma omp atomic
                            a[ch] += (double)v;
                        }
                    }
                    const float c_d = depths[gid];
                    const float c_u = (depths[gid] - gt_px_depth) * (depths[gid] - gt_px_depth);
                    accum_depth_rec = last_alpha * last_depth + (1.f - last_alpha) * accum_depth_rec;
                    accum_uncertainty_rec = last_alpha * last_uncertainty + (1.f - last_alpha) * accum_uncertainty_rec;
                    last_depth = c_d;
                    last_uncertainty = c_u;
                    dL_dalpha += (c_d - accum_depth_rec) * dL_depth;
                    dL_dalpha += (c_u - accum_uncertainty_rec) * dL_duncertainty;
                    if (a) {
                        const float v = dchannel_dcolor * dL_depth + 2. * (depths[gid] - gt_px_depth) * dchannel_dcolor * dL_duncertainty;
#pragma omp atomic
                        a[3] += (double)v;
                    }
                    ddepth_dalpha = T * (c_d - accum_depth_rec);
                    dL_dalpha *= T;
                    last_alpha = alpha;
                    float bg_dot_dpixel = 0;
                    for (int i = 0; i < 3; i++) bg_dot_dpixel += bg[i] * dL_dpixel[i];
                    dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot_dpixel;
                    const float dL_dG = co[3] * dL_dalpha;
                    const float gdx = G * dx, gdy = G * dy;
                    const float dG_ddelx = -gdx * co[0] - gdy * co[1];
                    const float dG_ddely = -gdy * co[2] - gdx * co[1];
                    if (pose_pass) {
                        PairRec r;
                        r.gid = gid;
                        r.dpix_dgc = dchannel_dcolor;
                        for (int ch = 0; ch < 3; ch++) {
                            r.dndcs[0][ch] = (dpixel_dalpha[ch] * co[3] * dG_ddelx * ddelx_dx);
                            r.dndcs[1][ch] = (dpixel_dalpha[ch] * co[3] * dG_ddely * ddely_dy);
                        }
                        r.ddepth_dndcs[0] = ddepth_dalpha * co[3] * dG_ddelx * ddelx_dx;
                        r.ddepth_dndcs[1] = ddepth_dalpha * co[3] * dG_ddely * ddely_dy;
                        pairs.push_back(r);
                    } else {
                        const float v4 = dL_dG * dG_ddelx * ddelx_dx, v5 = dL_dG * dG_ddely * ddely_dy;
                        const float v6 = -0.5f * gdx * dx * dL_dG, v7 = -0.5f * gdx * dy * dL_dG;
                        const float v8 = -0.5f * gdy * dy * dL_dG, v9 = G * dL_dalpha;
#pragma omp atomic
                        a[4] += (double)v4;
#pragma omp atomic
                        a[5] += (double)v5;
#pragma omp atomic
                        a[6] += (double)v6;
#pragma omp atomic
                        a[7] += (double)v7;
#pragma omp atomic
                        a[8] += (double)v8;
#pragma omp atomic
                        a[9] += (double)v9;
                    }
                }
                if (pose_pass) tile_pairs[ly * BLOCK_X + lx] = pairs;
            }
        if (!pose_pass) continue;
        // ---- ComputePG for this tile (F/cr/backward.cu:864-1337).
        // Well-defined semantics (default): every pixel consumes all of its recorded pairs in order.
        // emulate_dropout: the reference's threads with no valid contributor (or outside the image) return BEFORE
        // the block-wide loads (:875-878, 935-938) and never fill collected_id[thread_rank]; the survivors then
        // compare against whatever that __shared__ slot held.  Under the survey's CPU execution (__shared__ ->
        // static, blocks run one after another) the slot keeps the value an earlier block left there; that run is
        // what this flag reproduces.  On a GPU the outcome is undefined.
        static thread_local std::vector<int> dummy;
        const int L = (int)(r1 - r0);
        const int rounds = (L + 255) / 256;
        std::vector<size_t> vseq(256, 0);
        std::vector<char> shut(256, 0);
        std::vector<std::array<V3, 12>> dpv(256);
        std::vector<std::array<float, 12>> ddv(256);
        for (int t = 0; t < 256; t++)
            for (int sidx = 0; sidx < 12; sidx++) { dpv[t][sidx] = {0, 0, 0}; ddv[t][sidx] = 0; }
        const V3* dgc = reinterpret_cast<const V3*>(dgc_dCampos_or_null);
        auto consume = [&](int t, const PairRec& r) {
            const uint32_t g = r.gid;
            V3 zero = {0, 0, 0};
            V3 dp_dCx = dgc ? r.dpix_dgc * dgc[(size_t)g * 3 + 0] : zero;
            V3 dp_dCy = dgc ? r.dpix_dgc * dgc[(size_t)g * 3 + 1] : zero;
            V3 dp_dCz = dgc ? r.dpix_dgc * dgc[(size_t)g * 3 + 2] : zero;
            V3 part1[12];
            part1[0] = dp_dCx * (-view[12]); part1[1] = dp_dCx * (-view[13]); part1[2] = dp_dCx * (-view[14]);
            part1[3] = dp_dCy * (-view[12]); part1[4] = dp_dCy * (-view[13]); part1[5] = dp_dCy * (-view[14]);
            part1[6] = dp_dCz * (-view[12]); part1[7] = dp_dCz * (-view[13]); part1[8] = dp_dCz * (-view[14]);
            part1[9] = dp_dCx * (-view[0]) + dp_dCy * (-view[4]) + dp_dCz * (-view[8]);
            part1[10] = dp_dCx * (-view[1]) + dp_dCy * (-view[5]) + dp_dCz * (-view[9]);
            part1[11] = dp_dCx * (-view[2]) + dp_dCy * (-view[6]) + dp_dCz * (-view[10]);
            const float* J = &st.dgndcs_dview[24 * (size_t)g];
            const V3 nx = {r.dndcs[0][0], r.dndcs[0][1], r.dndcs[0][2]};
            const V3 ny = {r.dndcs[1][0], r.dndcs[1][1], r.dndcs[1][2]};
            const float* mg = means + 3 * (size_t)g;
            const float wc[4] = {mg[0], mg[1], mg[2], 1.0f};
            for (int sidx = 0; sidx < 12; sidx++) {
                const V3 p21 = J[2 * sidx] * nx + J[2 * sidx + 1] * ny;
                const float d21 = J[2 * sidx] * r.ddepth_dndcs[0] + J[2 * sidx + 1] * r.ddepth_dndcs[1];
                dpv[t][sidx] = dpv[t][sidx] + (part1[sidx] + p21);
                const float d1 = (sidx % 3 == 2) ? r.dpix_dgc * wc[sidx / 3] : 0.f;
                ddv[t][sidx] = d1 + d21;  // assigned (F/cr/backward.cu:1278-1289)
            }
        };
        if (!emulate_dropout) {
            for (int t = 0; t < 256; t++)
                for (const PairRec& r : tile_pairs[t]) consume(t, r);
        } else {
            int toDo = L;
            for (int i = 0; i < rounds; i++, toDo -= 256) {
                for (int t = 0; t < 256; t++) {  // loads by the threads that are still alive
                    if (tile_pairs[t].empty()) continue;  // returned early: !inside or length == 0
                    const int progress = i * 256 + t;
                    if ((uint32_t)(r0 + progress) < r1) g_stale_collected_id[t] = (int)st.point_list[r1 - progress - 1];
                }
                for (int t = 0; t < 256; t++) {
                    if (tile_pairs[t].empty()) continue;
                    for (int j = 0; j < std::min(256, toDo); j++) {
                        const int global_id = g_stale_collected_id[j];
                        if (shut[t] || vseq[t] >= tile_pairs[t].size()) break;
                        if ((uint32_t)global_id == tile_pairs[t][vseq[t]].gid) {
                            consume(t, tile_pairs[t][vseq[t]]);
                            if (vseq[t] == tile_pairs[t].size() - 1) { shut[t] = 1; continue; }
                            vseq[t]++;
                        }
                    }
                }
            }
        }
        static const int slot2entry[12] = {0, 1, 2, 4, 5, 6, 8, 9, 10, 12, 13, 14};
        for (int ly = 0; ly < BLOCK_Y; ly++)
            for (int lx = 0; lx < BLOCK_X; lx++) {
                const int t = ly * BLOCK_X + lx;
                if (tile_pairs[t].empty()) continue;
                const uint32_t px = tx * BLOCK_X + lx, py = ty * BLOCK_Y + ly;
                const size_t pix_id = (size_t)W * py + px;
                float dL_dpixel[3];
                for (int i = 0; i < 3; i++) dL_dpixel[i] = dL_dpixels[i * N + pix_id];
                const float dL_depth = dL_depths[pix_id];
                for (int sidx = 0; sidx < 12; sidx++) {
                    const float v = dL_dpixel[0] * dpv[t][sidx].x + dL_dpixel[1] * dpv[t][sidx].y + dL_dpixel[2] * dpv[t][sidx].z +
                                    dL_depth * ddv[t][sidx];
                    vsum[slot2entry[sidx]] += (double)v;
                }
            }
    }
    if (dL_dview_sum)
        for (int i = 0; i < 16; i++) dL_dview_sum[i] = vsum[i];
}

}  // namespace

// =================================================================== C entry points
// ---------------------------------------------------------------------------------------------------------
// Analysis aid (not part of the reference): how many (pixel block, Gaussian) pairs a blend kernel has to visit when a
// 16x16 tile is split into 8x8 quadrants or into 4x4 blocks.  Light-variant termination rules.  out[0..9]:
//  0 blended (pixel, Gaussian) pairs            1 instances blended by at least one pixel
//  2 (8x8, Gaussian) pairs with a blended pixel 3 (4x4, Gaussian) pairs with a blended pixel
//  4 (8x8, Gaussian) pairs a box-culled forward visits (box of alpha >= 15/255 meets the block, block not finished)
//  5 the same for 4x4 blocks
//  6 sum over quadrants of max over its four 4x4 blocks of [5]-type counts (iterations of a wave whose 16-lane rows
//    walk one block's list each)               7 the same for [3]-type counts (backward)
//  8 sum over tiles of max over 4 quadrants of [4]-type counts   9 the same for [2]
void pairStats(const State& st, double* out, double* out2, double* out3 = nullptr) {
    const int W = st.W, H = st.H;
    const int tiles = st.gx * st.gy;
    double acc[10] = {0};
    double acc3[4] = {0};  // half tiles: 0 valid (16 wide x 8 high, Gaussian) pairs, 1 box-tested; 2/3 the same for 8 wide x 16 high
    double acc2[8] = {0};  // 0 valid (8x4, Gaussian) pairs, 1 tested, 2/3 wave iterations bwd/fwd with per-128-batch max over the
                           // two halves, 4/5 the same for four 4x4 rows, 6/7 for the whole quadrant (one list per wave)
#pragma omp parallel for schedule(dynamic, 4)
    for (int tile = 0; tile < tiles; tile++) {
        const uint32_t r0 = st.ranges[2 * tile], r1 = st.ranges[2 * tile + 1];
        const int tx = tile % st.gx, ty = tile / st.gx;
        float T[256];
        bool done[256], inside[256];
        for (int p = 0; p < 256; p++) {
            const uint32_t px = tx * 16 + (p & 15), py = ty * 16 + (p >> 4);
            inside[p] = px < (uint32_t)W && py < (uint32_t)H;
            T[p] = 1.f;
            done[p] = !inside[p];
        }
        double loc[10] = {0};
        double loc2[8] = {0}, loc3[4] = {0};
        int hb_valid[8] = {0}, hb_tested[8] = {0}, bb_valid[16] = {0}, bb_tested[16] = {0}, qb_valid[4] = {0}, qb_tested[4] = {0};
        auto close_batch = [&]() {
            for (int q = 0; q < 4; q++) {
                const int qx = q & 1, qy = q >> 1;
                const int h0 = qy * 4 + qx, h1 = h0 + 2;  // halves: index = (y / 4) * 2 + x / 8
                loc2[2] += std::max(hb_valid[h0], hb_valid[h1]);
                loc2[3] += std::max(hb_tested[h0], hb_tested[h1]);
                int mv = 0, mt = 0;
                for (int r = 0; r < 4; r++) {
                    const int b = ((qy) * 2 + (r >> 1)) * 4 + qx * 2 + (r & 1);
                    mv = std::max(mv, bb_valid[b]);
                    mt = std::max(mt, bb_tested[b]);
                }
                loc2[4] += mv; loc2[5] += mt;
                loc2[6] += qb_valid[q]; loc2[7] += qb_tested[q];
            }
            for (int i = 0; i < 8; i++) hb_valid[i] = hb_tested[i] = 0;
            for (int i = 0; i < 16; i++) bb_valid[i] = bb_tested[i] = 0;
            for (int i = 0; i < 4; i++) qb_valid[i] = qb_tested[i] = 0;
        };
        int q_tested[4] = {0}, q_valid[4] = {0}, b_tested[16] = {0}, b_valid[16] = {0};
        for (uint32_t k = r0; k < r1; k++) {
            if (k > r0 && ((k - r0) % 128) == 0) close_batch();
            bool all_done = true;
            for (int p = 0; p < 256; p++) all_done = all_done && done[p];
            if (all_done) break;
            const uint32_t id = st.point_list[k];
            const float gx_ = st.means2D[2 * (size_t)id], gy_ = st.means2D[2 * (size_t)id + 1];
            const float* co = &st.conic_opacity[4 * (size_t)id];
            // box of the region alpha >= 15/255
            const float tau = 2.f * std::log(co[3] * 255.f / 15.f);
            const float det = co[0] * co[2] - co[1] * co[1];
            float hx = 1e9f, hy = 1e9f;
            if (det > 0 && tau > 0) { hx = std::sqrt(tau * co[2] / det); hy = std::sqrt(tau * co[0] / det); }
            const float lx = gx_ - tx * 16, ly = gy_ - ty * 16;
            bool blended[256];
            bool any = false;
            for (int p = 0; p < 256; p++) {
                blended[p] = false;
                if (done[p]) continue;
                const float pfx = (float)(tx * 16 + (p & 15)), pfy = (float)(ty * 16 + (p >> 4));
                const float dx = gx_ - pfx, dy = gy_ - pfy;
                const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                if (power > 0.0f) continue;
                const float alpha = std::min<decltype(co[3] * m_exp(power))>(0.99f, co[3] * m_exp(power));
                if (alpha < 15.0f / 255.0f) continue;
                const float test_T = T[p] * (1 - alpha);
                if (test_T < 0.0001f) { done[p] = true; continue; }
                T[p] = test_T;
                blended[p] = true;
                any = true;
                loc[0] += 1;
            }
            if (any) loc[1] += 1;
            for (int q = 0; q < 4; q++) {
                const int x0 = (q & 1) * 8, y0 = (q >> 1) * 8;
                bool v = false, alive = false;
                for (int yy = 0; yy < 8; yy++)
                    for (int xx = 0; xx < 8; xx++) {
                        const int p = (y0 + yy) * 16 + x0 + xx;
                        v = v || blended[p];
                        alive = alive || !done[p] || blended[p];
                    }
                const bool box = tau > 0 && lx + hx >= x0 && lx - hx <= x0 + 7 && ly + hy >= y0 && ly - hy <= y0 + 7;
                if (v) { loc[2] += 1; q_valid[q]++; qb_valid[q]++; }
                if (box && alive) { loc[4] += 1; q_tested[q]++; qb_tested[q]++; }
            }
            for (int b = 0; b < 16; b++) {
                const int x0 = (b & 3) * 4, y0 = (b >> 2) * 4;
                bool v = false, alive = false;
                for (int yy = 0; yy < 4; yy++)
                    for (int xx = 0; xx < 4; xx++) {
                        const int p = (y0 + yy) * 16 + x0 + xx;
                        v = v || blended[p];
                        alive = alive || !done[p] || blended[p];
                    }
                const bool box = tau > 0 && lx + hx >= x0 && lx - hx <= x0 + 3 && ly + hy >= y0 && ly - hy <= y0 + 3;
                if (v) { loc[3] += 1; b_valid[b]++; bb_valid[b]++; }
                if (box && alive) { loc[5] += 1; b_tested[b]++; bb_tested[b]++; }
            }
            for (int ht = 0; ht < 4; ht++) {  // half tiles: 0/1 top/bottom (16 x 8), 2/3 left/right (8 x 16)
                const int x0 = ht == 3 ? 8 : 0, y0 = ht == 1 ? 8 : 0, wx = ht < 2 ? 16 : 8, wy = ht < 2 ? 8 : 16;
                bool v = false, alive = false;
                for (int yy = 0; yy < wy; yy++)
                    for (int xx = 0; xx < wx; xx++) {
                        const int p = (y0 + yy) * 16 + x0 + xx;
                        v = v || blended[p];
                        alive = alive || !done[p] || blended[p];
                    }
                const bool box = tau > 0 && lx + hx >= x0 && lx - hx <= x0 + wx - 1 && ly + hy >= y0 && ly - hy <= y0 + wy - 1;
                if (v) loc3[ht < 2 ? 0 : 2] += 1;
                if (box && alive) loc3[ht < 2 ? 1 : 3] += 1;
            }
            for (int hb = 0; hb < 8; hb++) {  // 8 wide x 4 high halves of the quadrants
                const int x0 = (hb & 1) * 8, y0 = (hb >> 1) * 4;
                bool v = false, alive = false;
                for (int yy = 0; yy < 4; yy++)
                    for (int xx = 0; xx < 8; xx++) {
                        const int p = (y0 + yy) * 16 + x0 + xx;
                        v = v || blended[p];
                        alive = alive || !done[p] || blended[p];
                    }
                const bool box = tau > 0 && lx + hx >= x0 && lx - hx <= x0 + 7 && ly + hy >= y0 && ly - hy <= y0 + 3;
                if (v) { loc2[0] += 1; hb_valid[hb]++; }
                if (box && alive) { loc2[1] += 1; hb_tested[hb]++; }
            }
        }
        close_batch();
        for (int q = 0; q < 4; q++) {
            int mt = 0, mv = 0;
            for (int r = 0; r < 4; r++) {
                const int b = ((q >> 1) * 2 + (r >> 1)) * 4 + (q & 1) * 2 + (r & 1);
                mt = std::max(mt, b_tested[b]);
                mv = std::max(mv, b_valid[b]);
            }
            loc[6] += mt;
            loc[7] += mv;
        }
        loc[8] += std::max(std::max(q_tested[0], q_tested[1]), std::max(q_tested[2], q_tested[3]));
        loc[9] += std::max(std::max(q_valid[0], q_valid[1]), std::max(q_valid[2], q_valid[3]));
#pragma omp critical
        {
            for (int i = 0; i < 10; i++) acc[i] += loc[i];
            for (int i = 0; i < 8; i++) acc2[i] += loc2[i];
            for (int i = 0; i < 4; i++) acc3[i] += loc3[i];
        }
    }
    for (int i = 0; i < 10; i++) out[i] = acc[i];
    if (out2) for (int i = 0; i < 8; i++) out2[i] = acc2[i];
    if (out3) for (int i = 0; i < 4; i++) out3[i] = acc3[i];
}

extern "C" {
// y[i] = the exponential exactly as renderForward / renderBackward call it (m_exp on a float), so that the GPU's exp_p32 /
// exp_glibc (csrc/exact_math.h) can be compared with THIS library's binding bit for bit (tests/test_hip_exact_math.py)
void dgro_exp(const float* x, float* y, long n) {
    for (long i = 0; i < n; i++) y[i] = (float)m_exp(x[i]);
}
// ... and the two restated expf themselves, whichever binding this library was built with (tests/test_oracle_expf.py)
void dgro_expf_restated(const float* x, float* y, long n) {
    for (long i = 0; i < n; i++) y[i] = expf_restated(x[i]);
}
void dgro_expf_p32(const float* x, float* y, long n) {
    for (long i = 0; i < n; i++) y[i] = expf_p32(x[i]);
}
// which one the float build's blend loops call (0 = expf_p32, the default; 1 = expf_restated); returns the previous mode
int dgro_set_exp_mode(int mode) {
    const int old = g_exp_mode;
    g_exp_mode = mode ? 1 : 0;
    return old;
}
// expf_p32 against exp() in double on EVERY float whose bit pattern lies in [lo_bits, hi_bits] (negative floats: 0x80000000 =
// -0.0 ... 0xc2d00000 = -104.0).  out = {largest error in ulps where the result is normal, the same where it is denormal
// (in units of 2^-149), arguments whose result is not the correctly rounded one, arguments scanned, argument bits of the
// two maxima}.
void dgro_expf_p32_scan(uint32_t lo_bits, uint32_t hi_bits, double* out) {
    double worst_n = 0, worst_d = 0, at_n = 0, at_d = 0;
    long bad = 0, total = 0;
#pragma omp parallel
    {
        double ln = 0, ld = 0;
        uint32_t an = 0, ad = 0;
        long lb = 0, lt = 0;
#pragma omp for schedule(static)
        for (int64_t u = lo_bits; u <= (int64_t)hi_bits; u++) {
            const uint32_t ub = (uint32_t)u;
            float x;
            std::memcpy(&x, &ub, 4);
            const float y = expf_p32(x);
            const double t = ::exp((double)x);
            int e;
            std::frexp(t, &e);
            const double ulp = std::ldexp(1.0, std::max(e - 24, -149));
            const double err = std::fabs((double)y - t) / ulp;
            if (t >= 0x1p-126) {
                if (err > ln) { ln = err; an = ub; }
            } else if (err > ld) { ld = err; ad = ub; }
            lb += (y != (float)t);
            lt++;
        }
#pragma omp critical
        {
            if (ln > worst_n) { worst_n = ln; at_n = an; }
            if (ld > worst_d) { worst_d = ld; at_d = ad; }
            bad += lb;
            total += lt;
        }
    }
    out[0] = worst_n; out[1] = worst_d; out[2] = (double)bad; out[3] = (double)total; out[4] = at_n; out[5] = at_d;
}
void dgro_pair_stats(void* st, double* out) { pairStats(*(State*)st, out, nullptr); }
void dgro_pair_stats2(void* st, double* out, double* out2) { pairStats(*(State*)st, out, out2); }
void dgro_pair_stats3(void* st, double* out, double* out2, double* out3) { pairStats(*(State*)st, out, out2, out3); }

// Test infrastructure for the median depth (L/cr/forward.cu:353-364, L/cr/backward.cu:1545-1560): the forward takes the
// Gaussian at which the transmittance crosses 0.5, the backward finds it again from a transmittance it reconstructs by
// division.  Two correct implementations can disagree on that Gaussian when some T_k lies within rounding of 0.5 -- with
// no trace in any output image when the neighbours have (nearly) the same depth.  out[pixel] = min_k |T_k - 0.5| over the
// transmittances before / after every blended Gaussian of the pixel (forward order, fp32, same thresholds as
// renderForward<true>) and, with `alphas` (the alpha image the backward is given), over the backward's reconstructed
// sequence; the parity tests mask pixels whose margin is within the reconstruction error (tests/util.py).
void dgro_light_median_margin(void* sp, const float* alphas, float* out) {
    State& st = *static_cast<State*>(sp);
    const int W = st.W, H = st.H, tiles = st.gx * st.gy;
#pragma omp parallel for schedule(dynamic, 4)
    for (int tile = 0; tile < tiles; tile++) {
        const uint32_t r0 = st.ranges[2 * tile], r1 = st.ranges[2 * tile + 1];
        const int tx = tile % st.gx, ty = tile / st.gx;
        for (int ly = 0; ly < BLOCK_Y; ly++)
            for (int lx = 0; lx < BLOCK_X; lx++) {
                const uint32_t px = tx * BLOCK_X + lx, py = ty * BLOCK_Y + ly;
                if (!(px < (uint32_t)W && py < (uint32_t)H)) continue;
                const float pfx = (float)px, pfy = (float)py;
                float T = 1.0f, margin = 0.5f;
                std::vector<float> blended;
                for (uint32_t k = r0; k < r1; k++) {
                    const uint32_t id = st.point_list[k];
                    const float dx = st.means2D[2 * (size_t)id] - pfx, dy = st.means2D[2 * (size_t)id + 1] - pfy;
                    const float* co = &st.conic_opacity[4 * (size_t)id];
                    const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                    if (power > 0.0f) continue;
                    const float alpha = std::min<decltype(co[3] * m_exp(power))>(0.99f, co[3] * m_exp(power));
                    if (alpha < 15.0f / 255.0f) continue;
                    const float test_T = T * (1 - alpha);
                    if (test_T < 0.0001f) break;
                    margin = std::min(margin, std::fabs(test_T - 0.5f));
                    T = test_T;
                    blended.push_back(alpha);
                }
                // the backward's view of the same sequence: T_final = 1 - alpha image, then one division per Gaussian
                // (L/cr/backward.cu:477,1500) -- off from the forward's products by the rounding of the alpha sum
                // divided by T_final, which is why the margin has to be taken on THIS sequence
                if (alphas) {
                    float Tb = 1.0f - alphas[(size_t)W * py + px];
                    for (size_t j = blended.size(); j-- > 0;) {
                        Tb = Tb / (1.f - blended[j]);
                        margin = std::min(margin, std::fabs(Tb - 0.5f));
                    }
                }
                out[(size_t)W * py + px] = margin;
            }
    }
}


void* dgro_state_new() { return new State(); }
void dgro_state_free(void* s) { delete static_cast<State*>(s); }

#define FIELDS(X)                                                                                   \
    X(depths) X(means2D) X(cov3D) X(conic_opacity) X(rgb) X(clamped) X(radii) X(tiles_touched)      \
    X(point_offsets) X(keys_unsorted) X(keys) X(point_list_unsorted) X(point_list) X(ranges)        \
    X(n_contrib) X(n_valid_contrib) X(final_T) X(dgndcs_dview) X(dg_camd)

// Borrow a state array: returns element count, *ptr -> data, *elem -> element size in bytes.
long dgro_state_get(void* sp, const char* name, const void** ptr, int* elem) {
    State& st = *static_cast<State*>(sp);
#define X(f)                                     \
    if (!std::strcmp(name, #f)) {                \
        *ptr = st.f.data();                      \
        *elem = (int)sizeof(st.f[0]);            \
        return (long)st.f.size();                \
    }
    FIELDS(X)
#undef X
    return -1;
}
// Overwrite a state array (stage-wise checks with another implementation's upstream values).
int dgro_state_set(void* sp, const char* name, const void* src, long count) {
    State& st = *static_cast<State*>(sp);
#define X(f)                                                       \
    if (!std::strcmp(name, #f)) {                                  \
        st.f.resize(count);                                        \
        std::memcpy(st.f.data(), src, count * sizeof(st.f[0]));    \
        return 0;                                                  \
    }
    FIELDS(X)
#undef X
    return -1;
}
void dgro_state_set_dims(void* sp, int P, int W, int H) {
    State& st = *static_cast<State*>(sp);
    st.P = P; st.W = W; st.H = H;
    st.gx = (W + BLOCK_X - 1) / BLOCK_X;
    st.gy = (H + BLOCK_Y - 1) / BLOCK_Y;
}
int dgro_state_num_rendered(void* sp) { return static_cast<State*>(sp)->R; }

// L/cr/rasterizer_impl.cu:54-66,141-153
void dgro_mark_visible(int P, const float* means, const float* view, const float* proj, uint8_t* present) {
    (void)proj;
    for (int idx = 0; idx < P; idx++) {
        V3 p = {means[3 * idx], means[3 * idx + 1], means[3 * idx + 2]};
        present[idx] = !(transformPoint4x3(p, view).z <= 0.2f);
    }
}

int dgro_preprocess(void* sp, int P, int D, int M, int W, int H, const float* means3D, const float* shs,
                    const float* colors_precomp, const float* opacities, const float* scales, float scale_modifier,
                    const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                    const float* projmatrix, const float* campos, float tan_fovx, float tan_fovy, int prefiltered) {
    State& st = *static_cast<State*>(sp);
    const float focal_y = H / (2.0f * tan_fovy);
    const float focal_x = W / (2.0f * tan_fovx);
    return preprocessForward(st, P, D, M, means3D, scales, scale_modifier, rotations, opacities, shs, cov3D_precomp,
                             colors_precomp, viewmatrix, projmatrix, campos, W, H, tan_fovx, tan_fovy, focal_x,
                             focal_y, prefiltered);
}
int dgro_binning(void* sp) {
    State& st = *static_cast<State*>(sp);
    binning(st);
    return st.R;
}
void dgro_light_render_forward(void* sp, const float* bg, const float* colors_precomp, const float* gt_depth,
                               float* out_color, float* out_depth, float* out_median, float* out_alpha,
                               float* out_depth_var, float* gau_uncertainty, int32_t* gau_related_pixels) {
    State& st = *static_cast<State*>(sp);
    const float* feat = colors_precomp ? colors_precomp : st.rgb.data();
    renderForward<true>(st, feat, bg, gt_depth, out_color, out_depth, out_median, out_alpha, out_depth_var,
                        gau_uncertainty, gau_related_pixels);
}

// Mirror of CudaRasterizer::Rasterizer::forward, light (L/cr/rasterizer.h:40-70,
// L/cr/rasterizer_impl.cu:197-350).  Outputs must be zero-initialised by the caller as
// L/rasterize_points.cu:69-76 does.  Returns num_rendered, or -1 on prefiltered violation.
int dgro_light_forward(void* sp, int P, int D, int M, const float* background, int width, int height,
                       const float* means3D, const float* shs, const float* colors_precomp, const float* opacities,
                       const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
                       const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx,
                       float tan_fovy, int prefiltered, float* out_color, float* out_depth, float* out_median_depth,
                       float* out_alpha, const float* gt_depth, float* out_depth_var, float* gau_uncertainty,
                       int32_t* gau_related_pixels, int32_t* radii) {
    State& st = *static_cast<State*>(sp);
    if (dgro_preprocess(sp, P, D, M, width, height, means3D, shs, colors_precomp, opacities, scales, scale_modifier,
                        rotations, cov3D_precomp, viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy, prefiltered))
        return -1;
    if (radii) std::memcpy(radii, st.radii.data(), sizeof(int32_t) * P);
    binning(st);
    dgro_light_render_forward(sp, background, colors_precomp, gt_depth, out_color, out_depth, out_median_depth,
                              out_alpha, out_depth_var, gau_uncertainty, gau_related_pixels);
    return st.R;
}

// Mirror of CudaRasterizer::Rasterizer::backward, light (L/cr/rasterizer.h:72-104,
// L/cr/rasterizer_impl.cu:354-495) plus the reduction L/diff_gaussian_rasterization/__init__.py:160-161
// (dL_dview16 = sum over pixels, accumulated in double).  Gradient buffers must arrive zeroed
// (L/rasterize_points.cu:174-187).  dL_dview_pix ([H*W,16]) may be NULL.
void dgro_light_backward(void* sp, int P, int D, int M, const float* background, const float* means3D,
                         const float* shs, const float* colors_precomp, const float* alphas, const float* scales,
                         float scale_modifier, const float* rotations, const float* cov3D_precomp,
                         const float* viewmatrix, const float* projmatrix, const float* campos, float tan_fovx,
                         float tan_fovy, const float* dL_dpix, const float* dL_dpix_depth,
                         const float* dL_dpix_median_depth, const float* dL_dpix_depth_var, float* dL_dmean2D,
                         float* dL_dconic, float* dL_dopacity, float* dL_dcolor, float* dL_ddepth, float* dL_dmean3D,
                         float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot,
                         const float* perspec_matrix, float* dL_dview_pix, float* dL_dview16, const float* gt_depth,
                         int track_off, int map_off) {
    State& st = *static_cast<State*>(sp);
    const int W = st.W, H = st.H;
    const float focal_y = H / (2.0f * tan_fovy);
    const float focal_x = W / (2.0f * tan_fovx);
    const int32_t* radii = st.radii.data();
    if (!track_off) poseGradientPre(st, P, means3D, radii, projmatrix, perspec_matrix);
    const float* color_ptr = colors_precomp ? colors_precomp : st.rgb.data();
    std::vector<double> acc((size_t)P * 13, 0.0);
    double vsum[16];
    renderBackwardLight(st, background, color_ptr, alphas, dL_dpix, dL_dpix_depth, dL_dpix_median_depth,
                        dL_dpix_depth_var, means3D, viewmatrix, gt_depth, track_off, map_off, acc, dL_dview_pix, vsum);
    for (int i = 0; i < 16; i++) dL_dview16[i] = (float)vsum[i];
    for (int g = 0; g < P; g++) {
        const double* a = &acc[(size_t)g * 13];
        for (int c = 0; c < 3; c++) dL_dcolor[3 * (size_t)g + c] += (float)a[c];
        dL_ddepth[g] += (float)a[3];
        dL_dmean2D[3 * (size_t)g + 0] += (float)a[4];
        dL_dmean2D[3 * (size_t)g + 1] += (float)a[5];
        dL_dconic[4 * (size_t)g + 0] += (float)a[6];
        dL_dconic[4 * (size_t)g + 1] += (float)a[7];
        dL_dconic[4 * (size_t)g + 3] += (float)a[8];
        dL_dopacity[g] += (float)a[9];
        for (int c = 0; c < 3; c++) dL_dmean3D[3 * (size_t)g + c] += (float)a[10 + c];
    }
    if (!map_off) {
        const float* cov3D_ptr = cov3D_precomp ? cov3D_precomp : st.cov3D.data();
        cov2DBackwardLight(st, P, means3D, radii, cov3D_ptr, focal_x, focal_y, tan_fovx, tan_fovy, viewmatrix,
                           dL_dconic, dL_dmean3D, dL_dcov3D);
        preprocessBackwardLight(P, D, M, means3D, radii, shs, st.clamped.data(), scales, rotations, scale_modifier,
                                viewmatrix, projmatrix, campos, dL_dmean2D, dL_dmean3D, dL_dcolor, dL_ddepth,
                                dL_dcov3D, dL_dsh, dL_dscale, dL_drot);
    }
}


// Mirror of CudaRasterizer::Rasterizer::forward, full (F/cr/rasterizer.h:40-67, F/cr/rasterizer_impl.cu:349-500).
// Returns num_rendered (or -1); *num_related receives the total of n_valid_contrib (the second host read there).
int dgro_full_forward(void* sp, int P, int D, int M, const float* background, int width, int height,
                      const float* means3D, const float* shs, const float* colors_precomp, const float* opacities,
                      const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
                      const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx,
                      float tan_fovy, int prefiltered, float* out_color, float* out_depth, const float* gt_depth,
                      float* out_uncertainty, int32_t* radii, int* num_related) {
    State& st = *static_cast<State*>(sp);
    if (dgro_preprocess(sp, P, D, M, width, height, means3D, shs, colors_precomp, opacities, scales, scale_modifier,
                        rotations, cov3D_precomp, viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy, prefiltered))
        return -1;
    if (radii) std::memcpy(radii, st.radii.data(), sizeof(int32_t) * P);
    binning(st);
    const float* feat = colors_precomp ? colors_precomp : st.rgb.data();
    renderForward<false>(st, feat, background, gt_depth, out_color, out_depth, nullptr, out_uncertainty, nullptr, nullptr,
                         nullptr);
    long ng = 0;
    for (uint32_t v : st.n_valid_contrib) ng += v;
    if (num_related) *num_related = (int)ng;
    return st.R;
}

// Mirror of CudaRasterizer::Rasterizer::backward, full (F/cr/rasterizer.h:69-102, F/cr/rasterizer_impl.cu:504-666):
// blend backward, computeCov2DCUDA + preprocessCUDA, ComputePG.  Gradient buffers arrive zeroed
// (F/rasterize_points.cu:161-171); dL_dview16 is the [4,4] the binding returns.
void dgro_full_backward(void* sp, int P, int D, int M, const float* background, const float* means3D, const float* shs,
                        const float* colors_precomp, const float* scales, float scale_modifier, const float* rotations,
                        const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix, const float* campos,
                        float tan_fovx, float tan_fovy, const float* dL_dpix, const float* dL_depths, float* dL_dmean2D,
                        float* dL_dconic, float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D,
                        float* dL_dsh, float* dL_dscale, float* dL_drot, const float* perspec_matrix, float* dL_dview16,
                        float* dL_dgau_depth, const float* gt_depth, const float* dL_duncertainties, int emulate_dropout) {
    State& st = *static_cast<State*>(sp);
    const int W = st.W, H = st.H;
    const float focal_y = H / (2.0f * tan_fovy);
    const float focal_x = W / (2.0f * tan_fovx);
    const int32_t* radii = st.radii.data();
    const float* color_ptr = colors_precomp ? colors_precomp : st.rgb.data();
    std::vector<double> acc((size_t)P * 10, 0.0);
    renderBackwardFull(st, background, color_ptr, dL_dpix, dL_depths, dL_duncertainties, gt_depth, acc, means3D, viewmatrix,
                       nullptr, false, nullptr);
    for (int g = 0; g < P; g++) {
        const double* a = &acc[(size_t)g * 10];
        for (int c = 0; c < 3; c++) dL_dcolor[3 * (size_t)g + c] += (float)a[c];
        dL_dgau_depth[g] += (float)a[3];
        dL_dmean2D[3 * (size_t)g + 0] += (float)a[4];
        dL_dmean2D[3 * (size_t)g + 1] += (float)a[5];
        dL_dconic[4 * (size_t)g + 0] += (float)a[6];
        dL_dconic[4 * (size_t)g + 1] += (float)a[7];
        dL_dconic[4 * (size_t)g + 3] += (float)a[8];
        dL_dopacity[g] += (float)a[9];
    }
    const float* cov3D_ptr = cov3D_precomp ? cov3D_precomp : st.cov3D.data();
    cov2DBackwardLight(st, P, means3D, radii, cov3D_ptr, focal_x, focal_y, tan_fovx, tan_fovy, viewmatrix, dL_dconic,
                       dL_dmean3D, dL_dcov3D, dL_dgau_depth);
    std::vector<float> dgc((size_t)P * 9, 0.f);
    preprocessBackwardLight(P, D, M, means3D, radii, shs, st.clamped.data(), scales, rotations, scale_modifier, viewmatrix,
                            projmatrix, campos, dL_dmean2D, dL_dmean3D, dL_dcolor, nullptr, dL_dcov3D, dL_dsh, dL_dscale,
                            dL_drot, dgc.data());
    poseGradientPre(st, P, means3D, radii, projmatrix, perspec_matrix);  // dgndcs_dview, F/cr/backward.cu:516-534
    double vsum[16];
    renderBackwardFull(st, background, color_ptr, dL_dpix, dL_depths, dL_duncertainties, gt_depth, acc, means3D, viewmatrix,
                       dgc.data(), true, vsum, emulate_dropout != 0);
    for (int i = 0; i < 16; i++) dL_dview16[i] = (float)vsum[i];
}

}  // extern "C"
