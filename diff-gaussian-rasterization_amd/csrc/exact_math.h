// exact_math.h -- expf and float division with the CPU restatement's bits, for the blend kernels' alpha and transmittance (gfx950).
//
// Why: the light backward derives T_final = 1 - (alpha image) (L/cuda_rasterizer/backward.cu:477) and rebuilds every
// transmittance by dividing by (1 - alpha) (backward.cu:570).  Both amplify a last-bit difference of one alpha: by
// 1 / T_final on nearly opaque pixels, by alpha / (1 - alpha) <= 99 per division.  A blend kernel whose alpha differs from
// the restatement's in the last bit on some pairs therefore ends up 1e-5 .. 1e-4 away in the gradients although every
// single operation is accurate to an ulp.  The remedy is not more accuracy but the SAME bits -- of a function
// that is as faithful to the reference's `exp` (nvcc's expf, <= 2 ulp, which no CPU reproduces) as any other:
//
//  * exp_p32(x) (the default since round 8): an fp32-only expf.  k = round(x log2 e); r = x - k ln 2 in two fused steps
//    (Cody-Waite, ln 2 = hi + lo); e^r by a degree-6 polynomial whose first three coefficients are exactly 1, 1, 1/2
//    (inline constants) and whose other four minimise the relative error on |r| <= 0.3467 (3.6e-9); scaled by 2^k.
//    Every step is ONE IEEE single-precision operation (fused multiply-add, subtract, multiply) or an integer shift / add, so
//    oracle/dgr_oracle.cpp: expf_p32 -- the same operations with std::fmaf -- agrees bit for bit on any host.  Error against
//    exp() in double over ALL 1 120 927 745 floats of [-104, -0]: <= 0.892 ulp (0.858 where the result is denormal); 99.52 %
//    correctly rounded (tests/test_oracle_expf.py).
//  * exp_glibc(x) (alpha mode 2, kept for A/B): the algorithm glibc >= 2.27 uses for expf (ARM optimized routines'
//    exp2f-table form: one table of 32 doubles, a cubic in double, ~0.502 ulp) in the CDNA double pipe; equals
//    oracle/dgr_oracle.cpp: expf_restated bit for bit.  Rounds 5-7's default.  Supported: x <= 0.
//  * div_ref(a, b): correctly rounded a / b for normal operands without the scaling and fix-up steps of the compiler's
//    IEEE sequence: v_rcp_f32, quotient, exact residual, one correction.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dgr {

// 2^(i/32) as doubles, minus i << 47 in the bit pattern (so that adding k << 47 for k = 32 q + i yields 2^(k/32))
#define DGR_EXP2F_TABLE                                                                                             \
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull, 0x3fef72b83c7d517bull, \
    0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull, 0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, \
    0x3feedea64c123422ull, 0x3feece086061892dull, 0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, \
    0x3feea47eb03a5585ull, 0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull, \
    0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull, 0x3feee89f995ad3adull, \
    0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull, 0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, \
    0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull

__constant__ const uint64_t EXP2F_TABLE[32] = {DGR_EXP2F_TABLE};

// the workgroup's copy of the table (256 bytes of LDS): threads 0..31 fill it; the caller synchronises before use
__device__ __forceinline__ void exp_ref_table_fill(uint64_t* lds_tab, int tid) {
    if (tid < 32) lds_tab[tid] = EXP2F_TABLE[tid];
}

// expf(x) for x <= 0, bit for bit the function the CPU restatement evaluates (oracle/dgr_oracle.cpp: expf_restated --
// every operation below is an IEEE double operation, so the two agree on any machine; tests/test_hip_exact_math.py checks it,
// tests/test_oracle_expf.py checks the restatement against the host's libm).  `tab` = the LDS copy of EXP2F_TABLE.
//   The algorithm is glibc >= 2.27's expf: z = x N / ln 2 (N = 32); k = round(z), r = z - k; s = 2^(k / N) from the table;
//   y = s (1 + C2 r + C1 r^2 + C0 r^3) in double; return (float) y.  Same table, same cubic; the cubic by Horner's rule with
//   fused multiply-adds (glibc: (C0 r + C1) r^2 + (C2 r + 1), fused or not as its build decided), which moves y by ~1e-16
//   relative: the float result differs from a given libm's only where y lies that close to a rounding boundary of the float
//   grid -- about one argument in 2^28.
// CLAMP (the backward blend, which evaluates every lane of a listed pair without a pre-test): arguments below -104 -- a
// needle-shaped Gaussian seen from a pixel far off its axis reaches -1e4 -- are evaluated at -104, where the result is 0 as
// at every smaller argument (the restatement returns 0 below -103.97); unclamped, the exponent field of s wraps below -708.
// 13 vector instructions (10 of them in the double pipe) and one 8-byte LDS read; 52 cycles of issue per wave at 8 waves per
// SIMD against 14 for v_mul_f32 + v_exp_f32 (profiles/r5/exp_variants.txt).
template <bool CLAMP = false>
__device__ __forceinline__ float exp_glibc(float x, const uint64_t* tab) {
#pragma clang fp contract(off)  // z + SHIFT must round z first; the fused steps below are explicit
    if (CLAMP) x = fmaxf(x, -104.0f);
    constexpr double INVLN2N = 0x1.71547652b82fep+0 * 32.0, SHIFT = 0x1.8p+52;
    constexpr double C0 = 0x1.c6af84b912394p-5 / 32 / 32 / 32, C1 = 0x1.ebfce50fac4f3p-3 / 32 / 32, C2 = 0x1.62e42ff0c52d6p-1 / 32;
    const double z = INVLN2N * (double)x;
    const double kd0 = z + SHIFT;                       // round to nearest even; the integer k sits in the low mantissa bits
    const uint32_t ki = (uint32_t)__double2loint(kd0);
    const double r = z - (kd0 - SHIFT);                 // in [-1/2, 1/2]
    const uint2 t = reinterpret_cast<const uint2*>(tab)[ki & 31u];
    const double s = __hiloint2double((int)(t.y + (ki << 15)), (int)t.x);  // bits of 2^(i/32) + (k << 47): 2^(k/32)
    // (written out: a VOP3 instruction takes ONE scalar operand, so the Horner steps are ordered such that each needs one
    //  constant from an SGPR pair and only C1 sits in a VGPR pair; the compiler's choice was v_fmac_f64 with both C1 and C2
    //  copied into fresh VGPR pairs per evaluation -- two moves and two registers more inside 64-register kernels)
    double p, y;
    asm("v_fma_f64 %0, %2, %3, %4\n\t"
        "v_fma_f64 %0, %0, %2, %5\n\t"
        "v_fma_f64 %1, %0, %2, 1.0"
        : "=&v"(p), "=v"(y)
        : "v"(r), "s"(C0), "v"(C1), "s"(C2));
    return (float)(y * s);
}

// ---- the fp32-only expf (see the header): exactly these operations, in this order, in oracle/dgr_oracle.cpp: expf_p32.
//   t  = fma(x, log2 e, M),  M = 1.5 * 2^23 + 64      round-to-nearest-even integer k = t - M sits in t's low mantissa bits,
//   k  = t - M                                         already biased by 64 for the exponent step
//   r  = fma(k, -ln2_hi, x);  r = fma(k, -ln2_lo, r)   |r| <= 0.34658
//   p  = 1 + r (1 + r (1/2 + r (C3 + r (C4 + r (C5 + r C6)))))        six fused steps
//   2^(k+64) p by adding (bits of t) << 23 to the bits of p (the bits of M vanish in the shift); times 2^-64: exact for
//   normal results, ONE rounding for denormal ones.
// What each costs here (profiles/r8/exp_poly32.txt): v_fma / v_mul / v_sub_f32 with VGPR, inline-constant or LITERAL operands
// issue at 2.4 cycles per wave; v_max_f32, v_lshl_add_u32, v_rndne, v_cvt, v_ldexp -- like anything with an SGPR operand -- at
// 4.2; the double pipe at 5.2+.  Hence the magic shift instead of v_rndne + v_cvt, the exponent add instead of v_ldexp, and
// constants the compiler can encode as 32-bit literals of VOP2 forms (v_fmaak / v_fmamk / v_fmac): 11 fast instructions + 1
// (+ the clamp), measured 29.5 cycles per evaluation against 57.2 for exp_glibc in the same harness.
// CLAMP (the backward blend, which evaluates every lane of a listed pair without a pre-test): arguments below -104 -- a
// needle-shaped Gaussian seen from a pixel far off its axis reaches -1e4, where the exponent field would wrap -- are evaluated
// at -104, where the result is 0 as at every smaller argument (e^-104 < 2^-150); NaN too.  The restatement always clamps; an
// argument the forward's log-domain pre-test let through is >= -ln(255 o / 15) > -92 for every finite opacity.
// Supported: x <= 0 (a positive argument is a pair the callers reject before they look at the result).
#define DGR_EXP_LOG2E 0x1.715476p+0f
#define DGR_EXP_MAGIC 12582976.0f          /* 1.5 * 2^23 + 64 */
#define DGR_EXP_NLN2HI -0x1.62e430p-1f     /* -(float) ln 2 */
#define DGR_EXP_NLN2LO 0x1.05c610p-29f     /* -(ln 2 - (float) ln 2) */
#define DGR_EXP_C3 0x1.5554a4p-3f
#define DGR_EXP_C4 0x1.555688p-5f
#define DGR_EXP_C5 0x1.122faep-7f
#define DGR_EXP_C6 0x1.6b6e26p-10f
#define DGR_EXP_CLAMP -104.0f
#define DGR_EXP_UNBIAS 0x1p-64f
template <bool CLAMP = false>
__device__ __forceinline__ float exp_p32(float x) {
#pragma clang fp contract(off)  // every fused step below is explicit
    if (CLAMP) x = fmaxf(x, DGR_EXP_CLAMP);
    const float t = __builtin_fmaf(x, DGR_EXP_LOG2E, DGR_EXP_MAGIC);
    const float k = t - DGR_EXP_MAGIC;
    float r = __builtin_fmaf(k, DGR_EXP_NLN2HI, x);
    r = __builtin_fmaf(k, DGR_EXP_NLN2LO, r);
    float p = __builtin_fmaf(DGR_EXP_C6, r, DGR_EXP_C5);
    p = __builtin_fmaf(p, r, DGR_EXP_C4);
    p = __builtin_fmaf(p, r, DGR_EXP_C3);
    p = __builtin_fmaf(p, r, 0.5f);
    p = __builtin_fmaf(p, r, 1.0f);
    p = __builtin_fmaf(p, r, 1.0f);
    const uint32_t bits = (__builtin_bit_cast(uint32_t, t) << 23) + __builtin_bit_cast(uint32_t, p);
    return __builtin_bit_cast(float, bits) * DGR_EXP_UNBIAS;
}

// correctly rounded a / b for normal operands and quotient (no scaling, no fix-up of special cases): quotient from
// v_rcp_f32 (1 ulp), exact residual, one correction.  The correction's own rounding error is ~1e-7 of an ulp of the
// quotient, so the result is the correctly rounded one except when a / b lies that close to a rounding boundary.
__device__ __forceinline__ float div_ref(float a, float b, float& inv) {
    inv = __builtin_amdgcn_rcpf(b);
    const float q0 = a * inv;
    const float r = __builtin_fmaf(-b, q0, a);          // exact
    return __builtin_fmaf(r, inv, q0);
}

}  // namespace dgr
